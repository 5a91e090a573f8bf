#!/bin/bash
# r04_exp13.sh <tag> — round 4, batch 13 on ONE box (development tool): channels at their own centres at decimation 16 (kernel CHAN == 17):
# the bank's GPU tests, then timings beside the decimation-8 form.
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "filter_bank" > $O/gpu_tests_bank.txt 2>&1; rc=$?; echo "bank tests rc=$rc"; tail -6 $O/gpu_tests_bank.txt | cut -c1-300
[ $rc -eq 0 ] || exit 1
for spec in "8 28 255 8" "8 28 255 8 nco=0.0123" "8 28 255 8 nco=0.0123 tuning=1004096" "16 28 255 8 nco=0.0123"; do
  echo "# fbank_bench.py $spec" | tee -a $O/fbank.txt
  timeout -k 10 200 python3 tools/fbank_bench.py $spec 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-1500 | tee -a $O/fbank.txt | cut -c1-60
done
