#!/bin/bash
# r05_exp28.sh <tag> — round 5, batch 28 (development tool): the full-rate pipeline's last inverse pass with a group's refills issued AHEAD of its stores
# (libif_fir_ab_lf.so) instead of behind them (libif_fir_dev.so).
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
A=qo-100-tools_amd
export IF_FIR_DEBUG=1
LIBS="$A/libif_fir_dev.so $A/libif_fir_ab_lf.so"
for w in fir255_2p28 fir127_2p26 fir1023_2p28 fir255_dec5_2p28; do
  timeout -k 10 300 python3 tools/ab_inproc.py $w --libs $LIBS --rounds 10 --reps 40 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
done
