#!/bin/bash
# r04_exp8.sh <tag> — round 4, batch 8 on ONE box (development tool): GPU tests; compiler fences between the write and read phases
# of the LDS exchanges (libif_fir_ab_nofence.so = without) and the end game of the block queue (development bits 4096 / 8192: the
# next block is taken only after the current one is stored during the last 1 / 2 groups per workgroup), in-process A/B.
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1; echo "gpu tests rc=$?"; tail -5 $O/gpu_tests.txt | cut -c1-300
A=qo-100-tools_amd
for w in fir255_dec4_2p28 fir127_2p26 fir1023_2p28; do
  timeout -k 10 300 python3 tools/ab_inproc.py $w --libs $A/libif_fir_ab_nofence.so $A/libif_fir_dev.so --rounds 8 --reps 40 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_lds_fence.txt
done
for w in fir127_2p26 fir255_dec4_2p28 fir1023_2p28 fir255_2p28; do
  timeout -k 10 300 python3 tools/ab_inproc.py $w --variants 0 1004096 1008192 --rounds 10 --reps 40 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_endgame.txt
done
timeout -k 10 200 python3 tools/ab_inproc.py fir127_2p26 --variants 0 1004096 1008192 --rounds 8 --reps 60 --buffers 6 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_endgame.txt
for w in fir255_dec3_2p28 fir511_dec3_2p28; do
  timeout -k 10 300 python3 tools/ab_inproc.py $w --variants 0 3000 --rounds 8 --reps 30 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_odd.txt
done
