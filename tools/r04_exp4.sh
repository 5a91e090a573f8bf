#!/bin/bash
# r04_exp4.sh <tag> — round 4, batch 4 on ONE box (development tool): how many of a block's first / last rows keep the default cache
# policy (the rest are loaded `nt`): builds with IF_FIR_FFT_EDGE_MIN = 4 (= round 3), 8, 16, 64 (= no nt loads), on configs[1]
# (4 and 2 overlap rows, with the tail phase), configs[4] and the headline, interleaved in one process per workload.
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
L="qo-100-tools_amd/libif_fir_dev.so qo-100-tools_amd/libif_fir_ab_edge8.so qo-100-tools_amd/libif_fir_ab_edge16.so qo-100-tools_amd/libif_fir_ab_edge64.so"
timeout -k 10 400 python3 tools/ab_inproc.py fir127_2p26 --libs $L --variants 0 1003072 --rounds 10 --reps 60 2>&1 | grep -v amdgpu.ids | tee $O/ab_fir127_edge.txt
timeout -k 10 400 python3 tools/ab_inproc.py fir255_dec4_2p28 --libs $L --rounds 10 --reps 40 2>&1 | grep -v amdgpu.ids | tee $O/ab_headline_edge.txt
timeout -k 10 400 python3 tools/ab_inproc.py fir1023_2p28 --libs $L --rounds 8 --reps 30 2>&1 | grep -v amdgpu.ids | tee $O/ab_fir1023_edge.txt
