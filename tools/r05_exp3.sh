#!/bin/bash
# r05_exp3.sh <tag> — round 5, batch 3 on ONE box (development tool): GPU tests; filter-bank forms (gathers ahead of the products, table
# phasors, per-lane stores without a test per output) against round 4's library; the NCO kernels (table phasor) against round 4.
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
A=qo-100-tools_amd
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1; rc=$?; echo "gpu tests rc=$rc"; tail -5 $O/gpu_tests.txt | cut -c1-300
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python3 tools/fbank_ab.py --libs $A/libif_fir_ab_r4.so $A/libif_fir_dev.so --cases 16:8:freq 8:8:freq 64:8:freq 32:8:freq 4:8:freq 16:16:freq 8:16:freq 4:8:slots 8:3:slots 8:8:slots 16:16:slots 2>&1 | grep -v amdgpu.ids | tee $O/fbank_ab.txt
for w in fir255_dec4_2p28 fir255_2p28; do
  timeout -k 10 300 python3 tools/ab_inproc.py $w --nco 0.1234 --libs $A/libif_fir_ab_r4.so $A/libif_fir_dev.so --rounds 8 --reps 40 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_nco.txt
done
