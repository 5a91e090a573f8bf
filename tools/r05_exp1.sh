#!/bin/bash
# r05_exp1.sh <tag> — round 5, batch 1 on ONE box (development tool): GPU tests on the library with the opaque exchange read bases
# (lds_opaque) and in-process A/B against round 4's library (libif_fir_ab_r4.so = HEAD of round 4 built in a worktree).
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1; rc=$?; echo "gpu tests rc=$rc"; tail -5 $O/gpu_tests.txt | cut -c1-300
[ $rc -ne 0 ] && exit $rc
A=qo-100-tools_amd
for w in fir255_dec4_2p28 fir127_2p26 fir1023_2p28 fir255_dec3_2p28 fir255_2p28; do
  timeout -k 10 300 python3 tools/ab_inproc.py $w --libs $A/libif_fir_ab_r4.so $A/libif_fir_dev.so --rounds 10 --reps 40 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_lds_opaque.txt
done
timeout -k 10 300 python3 tools/ab_inproc.py fir255_dec4_2p28 --i16 --libs $A/libif_fir_ab_r4.so $A/libif_fir_dev.so --rounds 10 --reps 40 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_lds_opaque.txt
