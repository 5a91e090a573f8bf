#!/bin/bash
# r05_exp7.sh <tag> — round 5, batch 7 on ONE box (development tool): exchange-2 rounds interleaved with pass 2 (libif_fir_dev.so) and the small
# inverse's tables requested one exchange ahead (libif_fir_ab_pipe.so = + -DIF_FIR_FFT_INV_PIPE=1) against the committed library (libif_fir_ab_pf1.so).
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
A=qo-100-tools_amd
for w in fir255_dec4_2p28 fir127_2p26 fir255_2p28 fir1023_2p28 fir255_dec3_2p28 fir1023_dec8_2p28; do
  timeout -k 10 300 python3 tools/ab_inproc.py $w --libs $A/libif_fir_ab_pf1.so $A/libif_fir_dev.so $A/libif_fir_ab_pipe.so --rounds 10 --reps 40 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
done
timeout -k 10 300 python3 tools/ab_inproc.py fir255_dec4_2p28 --i16 --libs $A/libif_fir_ab_pf1.so $A/libif_fir_dev.so $A/libif_fir_ab_pipe.so --rounds 10 --reps 40 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
timeout -k 10 600 python3 tools/fbank_ab.py --libs $A/libif_fir_ab_pf1.so $A/libif_fir_dev.so $A/libif_fir_ab_pipe.so --cases 16:8:freq 8:8:freq 8:8:slots 4:8:slots 2>&1 | grep -v amdgpu.ids | tee $O/fbank_ab.txt
