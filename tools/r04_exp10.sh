#!/bin/bash
# r04_exp10.sh <tag> — round 4, batch 10 on ONE box (development tool): the decimation-8 bank's all-slots form (one launch per slot
# parity: two 8-point transforms per group give all eight slots of the parity): its GPU tests, then timings against the
# per-channel form (development launch 4096) at 4, 6, 8 and 16 channels.
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "filter_bank" > $O/gpu_tests_bank.txt 2>&1; rc=$?; echo "bank tests rc=$rc"; tail -6 $O/gpu_tests_bank.txt | cut -c1-300
[ $rc -eq 0 ] || exit 1
for spec in "8 28 255 8" "8 28 255 8 tuning=1004096" "16 28 255 8" "16 28 255 8 tuning=1004096" "4 28 255 8" "4 28 255 8 tuning=1004096" "6 28 255 8" "6 28 255 8 tuning=1004096" "8 28 1023 8" "16 28 1023 8"; do
  echo "== $spec" | tee -a $O/fbank.txt
  timeout -k 10 200 python3 tools/fbank_bench.py $spec 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-1500 | tee -a $O/fbank.txt | cut -c1-260
done
