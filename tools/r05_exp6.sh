#!/bin/bash
# r05_exp6.sh <tag> — round 5, batch 6 on ONE box (development tool): GPU tests on the library with the table prefetches everywhere they fit;
# A/B against round 4's library (libif_fir_ab_r4.so) for every workload family, in one process.
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
A=qo-100-tools_amd
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1; rc=$?; echo "gpu tests rc=$rc"; tail -5 $O/gpu_tests.txt | cut -c1-300
[ $rc -ne 0 ] && exit $rc
for w in fir255_dec4_2p28 fir127_2p26 fir255_2p28 fir1023_2p28 fir255_dec3_2p28 fir255_dec2_2p28 fir1023_dec8_2p28 fir2047_dec8_2p26 fir511_dec3_2p28; do
  timeout -k 10 300 python3 tools/ab_inproc.py $w --libs $A/libif_fir_ab_r4.so $A/libif_fir_dev.so --rounds 10 --reps 40 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_prefetch.txt
done
timeout -k 10 300 python3 tools/ab_inproc.py fir255_dec4_2p28 --i16 --libs $A/libif_fir_ab_r4.so $A/libif_fir_dev.so --rounds 10 --reps 40 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_prefetch.txt
timeout -k 10 300 python3 tools/ab_inproc.py fir255_dec4_2p28 --nco 0.1234 --libs $A/libif_fir_ab_r4.so $A/libif_fir_dev.so --rounds 10 --reps 40 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_prefetch.txt
timeout -k 10 900 python3 tools/fbank_ab.py --libs $A/libif_fir_ab_r4.so $A/libif_fir_dev.so --cases 16:8:freq 8:8:freq 64:8:freq 4:8:freq 8:8:slots 8:16:slots 16:16:slots 4:8:slots 2>&1 | grep -v amdgpu.ids | tee $O/fbank_ab.txt
