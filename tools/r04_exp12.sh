#!/bin/bash
# r04_exp12.sh <tag> — round 4, batch 12 on ONE box (development tool): the filter bank timed SETTLED (tools/fbank_bench.py now runs ~150 ms
# of its own launches ahead of the timed ones): every form of profiles/r04_filter_bank.txt again.
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
for spec in "8 28 255 4" "8 28 255 8" "8 28 255 8 tuning=1004096" "16 28 255 8" "16 28 255 8 tuning=1004096" "4 28 255 8" "4 28 255 8 tuning=1004096" "6 28 255 8" "6 28 255 8 tuning=1004096" "8 28 1023 8" "16 28 1023 8" "8 28 255 8 freq" "16 28 255 8 freq" "16 28 255 16" "8 28 255 16"; do
  echo "# fbank_bench.py $spec" | tee -a $O/fbank.txt
  timeout -k 10 200 python3 tools/fbank_bench.py $spec 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-1500 | tee -a $O/fbank.txt | cut -c1-60
done
