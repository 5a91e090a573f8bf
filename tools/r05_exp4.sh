#!/bin/bash
# r05_exp4.sh <tag> — round 5, batch 4 on ONE box (development tool): bank GPU tests; the filter-bank general forms against round 4's library and
# a build with batches of 8 terms (libif_fir_ab_nb8.so).
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
A=qo-100-tools_amd
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu -k "bank or channel" > $O/gpu_tests_bank.txt 2>&1; rc=$?; echo "bank gpu tests rc=$rc"; tail -3 $O/gpu_tests_bank.txt | cut -c1-300
[ $rc -ne 0 ] && exit $rc
timeout -k 10 900 python3 tools/fbank_ab.py --libs $A/libif_fir_ab_r4.so $A/libif_fir_dev.so $A/libif_fir_ab_nb8.so --cases 16:8:freq 8:8:freq 64:8:freq 4:8:freq 16:16:freq 8:8:slots 16:16:slots 8:3:slots 2>&1 | grep -v amdgpu.ids | tee $O/fbank_ab.txt
