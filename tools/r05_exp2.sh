#!/bin/bash
# r05_exp2.sh <tag> — round 5, batch 2 on ONE box (development tool): the filter-bank forms with their table gathers requested ahead
# of the products (gather_mac), against round 4's library, in one process; then the bank's GPU tests.
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
A=qo-100-tools_amd
timeout -k 10 600 python3 tools/fbank_ab.py --libs $A/libif_fir_ab_r4.so $A/libif_fir_dev.so --cases 16:8:freq 8:8:freq 64:8:freq 4:8:freq 16:16:freq 8:16:freq 4:8:slots 8:3:slots 2>&1 | grep -v amdgpu.ids | tee $O/fbank_ab.txt
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu -k "bank or channel" > $O/gpu_tests_bank.txt 2>&1; echo "bank gpu tests rc=$?"; tail -3 $O/gpu_tests_bank.txt | cut -c1-300
