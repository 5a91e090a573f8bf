#!/bin/bash
# r05_exp8.sh <tag> — round 5, batch 8 on ONE box (development tool): the cold load path drained before it joins the steady state (so that pass 1 of
# the steady state no longer waits for the previous block's stores: libif_fir_dev.so), the same with all four batches of next-block rows requested
# during pass 3 (libif_fir_ab_eg4.so), against the committed library (libif_fir_ab_pf1.so).
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
A=qo-100-tools_amd
for w in fir255_dec4_2p28 fir127_2p26 fir255_2p28 fir1023_2p28 fir255_dec3_2p28 fir1023_dec8_2p28 fir255_dec2_2p28; do
  timeout -k 10 300 python3 tools/ab_inproc.py $w --libs $A/libif_fir_ab_pf1.so $A/libif_fir_dev.so $A/libif_fir_ab_eg4.so --rounds 10 --reps 40 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
done
timeout -k 10 300 python3 tools/ab_inproc.py fir255_dec4_2p28 --i16 --libs $A/libif_fir_ab_pf1.so $A/libif_fir_dev.so $A/libif_fir_ab_eg4.so --rounds 10 --reps 40 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
timeout -k 10 600 python3 tools/fbank_ab.py --libs $A/libif_fir_ab_pf1.so $A/libif_fir_dev.so $A/libif_fir_ab_eg4.so --cases 16:8:freq 8:8:freq 8:8:slots 16:16:slots 2>&1 | grep -v amdgpu.ids | tee $O/fbank_ab.txt
