#!/bin/bash
# soak.sh <tag> [seeds...] — the random-configuration GPU test under other seeds (every kernel family, ragged pieces, formats,
# NCO, decimations), plus the power kernel's rate (development tool)
cd "$(dirname "$0")/.."
O=gpurun_out/$1; shift
mkdir -p $O
for seed in "$@"; do
  IF_FIR_TEST_SEED=$seed timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "random_configurations or random_filter_bank" 2>&1 | tail -1 | sed "s/^/seed $seed: /" | tee -a $O/soak.txt
done
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "power" 2>&1 | tail -1
python3 tools/power_bench.py 2>&1 | tail -1 | tee $O/power.txt
