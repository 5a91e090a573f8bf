#!/bin/bash
# r03_exp4.sh <tag> — round-3 batch 4 on ONE box: GPU tests with the tail phase of the block queue, then tail on / off
# (tuning 1000256 = tail off) interleaved on the three single-GPU BASELINE workloads, with per-wave run times.
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
export IF_FIR_DEBUG=1
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu 2>&1 | tee $O/pytest.txt
grep -q " passed" $O/pytest.txt && ! grep -q "failed\|error" $O/pytest.txt || exit 1
for w in fir255_dec4_2p28 fir127_2p26 fir1023_2p28 fir255_2p28; do
  timeout -k 10 300 python3 tools/sweep.py $w 100 1000256 100 1000256 100 1000256 100 1000256 > $O/sweep_$w.txt 2>&1; cut -c1-150 $O/sweep_$w.txt | grep variant
  timeout -k 10 300 python3 tools/fft_clock.py $w 0 1000256 0 1000256 > $O/clock_$w.txt 2>&1; grep variant $O/clock_$w.txt
done
