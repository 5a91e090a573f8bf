#!/bin/bash
# r04_exp3.sh <tag> — round 4, batch 3 on ONE box (development tool): GPU tests; configs[1] (127 taps, 2^26 = 17 477 blocks of 3840)
# on grids of 240..256 workgroups (a grid on which the 9 rounds divide evenly: 243 workgroups x 8 waves x 9 blocks) and on the
# 2-row kernel with 2 or 4 cached edge rows; the same for 2^27 and 2^25-sample calls.
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.txt
timeout -k 10 300 python3 tools/ab_inproc.py fir127_2p26 --variants 0 2243 2240 2246 2250 2235 1001024 --rounds 12 --reps 60 2>&1 | grep -v amdgpu.ids | tee $O/ab_fir127_grid.txt
timeout -k 10 300 python3 tools/ab_inproc.py fir127_2p26 --libs qo-100-tools_amd/libif_fir_ab_edge4.so --variants 0 1001024 1003072 --rounds 12 --reps 60 2>&1 | grep -v amdgpu.ids | tee $O/ab_fir127_edge4.txt
timeout -k 10 300 python3 tools/ab_inproc.py fir255_dec4_2p28 --variants 0 2250 2253 2248 --rounds 10 --reps 40 2>&1 | grep -v amdgpu.ids | tee $O/ab_headline_grid.txt
