#!/bin/bash
# r05_exp5.sh <tag> — round 5, batch 5 on ONE box (development tool): table reads of the transforms requested ahead of their butterflies
# (IF_FIR_FFT_TW_PREFETCH=1: libif_fir_ab_twpf.so) against the default build, in one process.
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
A=qo-100-tools_amd
for w in fir255_dec4_2p28 fir127_2p26 fir255_2p28 fir1023_2p28 fir255_dec3_2p28 fir255_dec2_2p28; do
  timeout -k 10 300 python3 tools/ab_inproc.py $w --libs $A/libif_fir_dev.so $A/libif_fir_ab_twpf.so --rounds 10 --reps 40 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_twpf.txt
done
timeout -k 10 300 python3 tools/ab_inproc.py fir255_dec4_2p28 --i16 --libs $A/libif_fir_dev.so $A/libif_fir_ab_twpf.so --rounds 10 --reps 40 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_twpf.txt
timeout -k 10 600 python3 tools/fbank_ab.py --libs $A/libif_fir_dev.so $A/libif_fir_ab_twpf.so --cases 16:8:freq 8:8:freq 8:8:slots 16:16:slots 4:8:slots 2>&1 | grep -v amdgpu.ids | tee $O/fbank_ab_twpf.txt
