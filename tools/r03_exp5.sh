#!/bin/bash
# r03_exp5.sh <tag> — round-3 batch 5 on ONE box: GPU tests (filter bank at decimation 8), filter-bank benchmark at 4 / 8 / 16
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
export IF_FIR_DEBUG=1
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu 2>&1 | tee $O/pytest.txt
grep -q " passed" $O/pytest.txt && ! grep -q "failed\|error" $O/pytest.txt || exit 1
for spec in "8 28 255 4" "8 28 255 8" "16 28 255 8" "8 28 255 16" "16 28 255 16" "16 28 1023 16"; do
  timeout -k 10 300 python3 tools/fbank_bench.py $spec 2>&1 | tail -1 | tee -a $O/fbank.txt
done
