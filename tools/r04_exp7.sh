#!/bin/bash
# r04_exp7.sh <tag> — round 4, batch 7 on ONE box (development tool): the odd-decimation kernel (decimation 3, 9, 15, ...): its GPU
# tests first (bounded), then the whole GPU suite, then its time against the selecting store (development variant 3000) and the
# filter bank's two forms at decimation 8.
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "odd_decimation or any_decimation" > $O/gpu_tests_odd.txt 2>&1; rc=$?; echo "odd tests rc=$rc"; tail -15 $O/gpu_tests_odd.txt | cut -c1-300
[ $rc -eq 0 ] || exit 1
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1; echo "gpu tests rc=$?"; tail -5 $O/gpu_tests.txt | cut -c1-300
for w in fir255_dec3_2p28 fir255_dec9_2p28 fir1023_dec3_2p28; do
  timeout -k 10 300 python3 tools/ab_inproc.py $w --variants 0 3000 --rounds 8 --reps 30 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_odd.txt
done
for spec in "8 28 255 8" "16 28 255 8"; do timeout -k 10 200 python3 tools/fbank_bench.py $spec 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-700 | tee -a $O/fbank.txt; done
A=qo-100-tools_amd
for w in fir127_2p26 fir255_2p28 fir1023_2p28 fir255_dec4_2p28; do
  timeout -k 10 300 python3 tools/ab_inproc.py $w --libs $A/libif_fir_ab_notan.so $A/libif_fir_dev.so --rounds 8 --reps 30 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_tan_full_rate.txt
done
