#!/bin/bash
# r05_exp21.sh <tag> — round 5, batch 21 (development tool): is it the delay or the deferral?  Tuning 1016384: the second wave of every SIMD requests its first block
# behind the table copy and the workgroup barrier, no sleep; 1081920: 1 us later still; 0: every wave requests its first block ahead of the table copy.
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
export IF_FIR_DEBUG=1
V="0 1016384 1081920"
for w in fir127_2p26 fir255_dec4_2p24 fir255_dec4_2p28 fir255_2p28 fir255_dec2_2p28 fir1023_dec8_2p28; do
  timeout -k 10 400 python3 tools/ab_inproc.py $w --variants $V --rounds 10 --reps 40 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
done
timeout -k 10 400 python3 tools/ab_inproc.py fir255_dec4_2p28 --i16 --variants $V --rounds 10 --reps 40 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
