#!/bin/bash
# r05_exp15.sh <tag> — round 5, batch 15 (development tool): the decimate-by-4 tail's next-block refills in chunks of four between a group's multiply-accumulate
# chains (libif_fir_ab_lc.so) instead of sixteen behind the group (libif_fir_dev.so): does a wave stall less at the issue of its loads?
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
A=qo-100-tools_amd
export IF_FIR_DEBUG=1
LIBS="$A/libif_fir_dev.so $A/libif_fir_ab_lc.so"
for w in fir255_dec4_2p28 fir1023_dec8_2p28 fir2047_dec8_2p26; do
  timeout -k 10 300 python3 tools/ab_inproc.py $w --libs $LIBS --rounds 10 --reps 40 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
done
timeout -k 10 300 python3 tools/ab_inproc.py fir255_dec4_2p28 --i16 --libs $LIBS --rounds 10 --reps 40 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
timeout -k 10 300 python3 tools/ab_inproc.py fir255_dec4_2p28 --nco 0.01 --libs $LIBS --rounds 10 --reps 40 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
