#!/bin/bash
# r04_exp5.sh <tag> — round 4, batch 5 on ONE box (development tool): GPU tests (per-channel centre frequencies of the decimation-8
# bank are new), then the number of cached edge rows of the full-rate pipeline (IF_FIR_FFT_EDGE_MIN_FULL = 4, 12, 16, 20, 24, 32)
# on configs[1] (4 / 2 overlap rows + tail), 255 taps D = 1 at 2^28, configs[4], and the bank's timing.
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1; echo "gpu tests rc=$?"; tail -12 $O/gpu_tests.txt | cut -c1-300
A=qo-100-tools_amd
L="$A/libif_fir_ab_full4.so $A/libif_fir_ab_full12.so $A/libif_fir_dev.so $A/libif_fir_ab_full20.so $A/libif_fir_ab_full24.so $A/libif_fir_ab_full32.so"
timeout -k 10 400 python3 tools/ab_inproc.py fir127_2p26 --libs $L --variants 0 1003072 --rounds 10 --reps 60 2>&1 | grep -v amdgpu.ids | tee $O/ab_fir127_edge.txt
timeout -k 10 400 python3 tools/ab_inproc.py fir255_2p28 --libs $L --rounds 8 --reps 30 2>&1 | grep -v amdgpu.ids | tee $O/ab_fir255_edge.txt
timeout -k 10 400 python3 tools/ab_inproc.py fir1023_2p28 --libs $A/libif_fir_dev.so $A/libif_fir_ab_full24.so $A/libif_fir_ab_full32.so --rounds 8 --reps 30 2>&1 | grep -v amdgpu.ids | tee $O/ab_fir1023_edge.txt
timeout -k 10 400 python3 tools/ab_inproc.py fir1023_dec8_2p28 --libs $A/libif_fir_ab_full4.so $A/libif_fir_dev.so --rounds 6 --reps 30 2>&1 | grep -v amdgpu.ids | tee $O/ab_dec8.txt
for spec in "8 28 255 8" "16 28 255 8" "8 28 255 16"; do timeout -k 10 200 python3 tools/fbank_bench.py $spec 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-600 | tee -a $O/fbank.txt; done
