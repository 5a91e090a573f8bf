#!/bin/bash
# r05_exp29.sh <tag> — round 5, batch 29 (development tool): the end of a short launch once more.  configs[1] is 8 two-wave rounds + 1093 blocks: more than one block
# per SIMD, so it has no tail phase (if_fir_fft_queue.h).  Tuning 1067584 (65536 + 2048): one block per SIMD goes to the tail anyway, the other 69 into the groups.
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
export IF_FIR_DEBUG=1
timeout -k 10 400 python3 tools/ab_inproc.py fir127_2p26 --variants 0 1067584 1002048 --rounds 10 --reps 40 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
timeout -k 10 400 python3 tools/ab_inproc.py fir127_2p26 --variants 0 1067584 --rounds 10 --reps 40 --buffers 6 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
timeout -k 10 400 python3 tools/ab_inproc.py fir255_dec4_2p24 --variants 0 1067584 --rounds 10 --reps 100 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
