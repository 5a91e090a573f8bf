#!/bin/bash
# r05_exp13.sh <tag> — round 5, batch 13 on ONE box (development tool): the decimate-by-2 tail's 16 entries of H per group in rolling batches (4 or 2 requested
# ahead, the next batch ahead of the previous one's products), with the compiler's pairs (d2p4) or with single reads (d2s4, d2s2), and the 4-point stage's
# twiddles of its small inverses requested ahead on top (d2p4i, d2s4i); against the library (libif_fir_dev.so: reads where they are used, paired).
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
A=qo-100-tools_amd
export IF_FIR_DEBUG=1
LIBS="$A/libif_fir_dev.so $A/libif_fir_ab_d2p4.so $A/libif_fir_ab_d2s4.so $A/libif_fir_ab_d2s4i.so $A/libif_fir_ab_d2p4i.so $A/libif_fir_ab_d2s2.so"
timeout -k 10 400 python3 tools/ab_inproc.py fir255_dec2_2p28 --libs $LIBS --rounds 10 --reps 40 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
timeout -k 10 400 python3 tools/ab_inproc.py fir255_dec2_2p28 --i16 --libs $LIBS --rounds 8 --reps 40 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
timeout -k 10 400 python3 tools/ab_inproc.py fir255_dec2_2p28 --nco 0.01 --libs $LIBS --rounds 8 --reps 40 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
