#!/bin/bash
# r05_exp20.sh <tag> — round 5, batch 20 (development tool): the head stagger in 1 us units (tuning 1000000 + 16384 + (k << 16): the second wave of every SIMD
# requests its first block k us late), k = 1..6, on short and long launches of both pipelines.
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
export IF_FIR_DEBUG=1
V="0 1081920 1147456 1212992 1278528 1344064 1409600"
for w in fir127_2p26 fir255_dec4_2p24 fir255_dec4_2p28 fir1023_2p28; do
  timeout -k 10 400 python3 tools/ab_inproc.py $w --variants $V --rounds 8 --reps 40 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
done
timeout -k 10 400 python3 tools/ab_inproc.py fir127_2p26 --variants $V --rounds 8 --reps 40 --buffers 6 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
