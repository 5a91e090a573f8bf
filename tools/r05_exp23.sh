#!/bin/bash
# r05_exp23.sh <tag> — round 5, batch 23 (development tool): the head of a launch, second step.  0 = the library (waves 0-3 request their rows ahead of the table
# copy, all eight waves copy); 1032768 = waves 4-7 alone copy the tables (in waves 0-3 the table loads return behind the 64 row loads, and the barrier with them);
# 1016384 = all eight request ahead (the form up to round 5).  Then the small calls, the GPU tests.
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
export IF_FIR_DEBUG=1
V="0 1032768 1016384"
for w in fir127_2p26 fir255_dec4_2p24 fir255_dec4_2p28 fir255_2p28 fir1023_2p28; do
  timeout -k 10 400 python3 tools/ab_inproc.py $w --variants $V --rounds 10 --reps 40 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
done
timeout -k 10 400 python3 tools/ab_inproc.py fir127_2p26 --variants $V --rounds 10 --reps 40 --buffers 6 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
timeout -k 10 200 python3 tools/small_calls.py 2>&1 | tail -6 | tee $O/small_calls.txt
