#!/bin/bash
# r05_exp19.sh <tag> — round 5, batch 19 (development tool): the head of a launch.  All waves request their first block together (67 MB in one burst), then all
# compute while the memory idles.  Tuning 1000000 + 16384 + (k << 16): the second wave of every SIMD requests its first block (k + 1) x 4 us late.
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
export IF_FIR_DEBUG=1
V="0 1016384 1081920 1147456 1212992"
timeout -k 10 400 python3 tools/ab_inproc.py fir127_2p26 --variants $V --rounds 10 --reps 40 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
timeout -k 10 400 python3 tools/ab_inproc.py fir127_2p26 --variants $V --rounds 10 --reps 40 --buffers 6 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
timeout -k 10 400 python3 tools/ab_inproc.py fir255_dec4_2p28 --variants $V --rounds 8 --reps 40 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
timeout -k 10 400 python3 tools/ab_inproc.py fir255_2p28 --variants $V --rounds 8 --reps 40 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
timeout -k 10 400 python3 tools/ab_inproc.py fir255_dec4_2p24 --variants $V --rounds 8 --reps 100 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
