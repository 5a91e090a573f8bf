#!/bin/bash
# r04_exp1.sh <tag> — round 4, batch 1 on ONE box (development tool): GPU tests, then configs[1] (127 taps, 2^26) on the 2-row
# kernel (L = 3968) against the 4-row kernel of round 3 (development variant 1024), each with and without the queue's tail
# phase extended to 16 two-wave rounds (variant 2048), interleaved.
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.txt
export IF_FIR_DEBUG=1
for r in 1 2 3; do
  timeout -k 10 200 python3 tools/sweep.py fir127_2p26 100 1001024 1002048 1003072 100 1001024 1002048 1003072 2>&1 | grep variant | cut -c1-120 | tee -a $O/fir127_rows_ab.txt
done
timeout -k 10 120 python3 tools/fft_clock.py fir127_2p26 0 1001024 1002048 2>&1 | tail -4 | cut -c1-250 | tee -a $O/fir127_rows_ab.txt
