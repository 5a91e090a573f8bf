#!/bin/bash
# r05_exp30.sh <tag> — round 5, batch 30 (development tool): single-round launches (a call of at most one block per wave of the chip is dealt one block per wave, slot-major
# over all CUs) against the form before (tuning 1262144: eight blocks per workgroup on an eighth of the CUs): time per back-to-back small call; the GPU tests first.
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
export IF_FIR_DEBUG=1
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -2 | tee $O/pytest.txt
grep -q " passed" $O/pytest.txt || exit 1
for t in 0 1262144 0 1262144; do
  timeout -k 10 200 python3 tools/small_calls.py $t 2>&1 | grep -v amdgpu.ids | tee -a $O/small_calls.txt
done
timeout -k 10 200 python3 tools/small_calls.py 0 1 127 2>&1 | grep -v amdgpu.ids | tee -a $O/small_calls.txt
timeout -k 10 200 python3 tools/small_calls.py 1262144 1 127 2>&1 | grep -v amdgpu.ids | tee -a $O/small_calls.txt
timeout -k 10 300 python3 tools/ab_inproc.py fir255_dec4_2p24 --variants 0 1262144 --rounds 8 --reps 100 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
timeout -k 10 300 python3 tools/ab_inproc.py fir255_dec4_2p28 --variants 0 1262144 --rounds 6 --reps 40 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
