#!/bin/bash
# r04_exp2.sh <tag> — round 4, batch 2 on ONE box (development tool): GPU tests with the (cos, tan) decimate-by-4 kernels, then
# in-process A/B timing (tools/ab_inproc.py): the headline with twiddles in (cos, tan) form against round 3's form
# (libif_fir_ab_notan.so = the same sources with -DIF_FIR_FFT_TAN=0), and configs[1] on 2 / 4 overlap rows with / without the
# queue's tail phase.
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1; echo "gpu tests rc=$?"; tail -5 $O/gpu_tests.txt
timeout -k 10 300 python3 tools/ab_inproc.py fir255_dec4_2p28 --libs qo-100-tools_amd/libif_fir_ab_notan.so qo-100-tools_amd/libif_fir_dev.so --rounds 16 --reps 40 2>&1 | grep -v amdgpu.ids | tee $O/ab_tan_headline.txt
timeout -k 10 300 python3 tools/ab_inproc.py fir255_dec4_2p28 --i16 --libs qo-100-tools_amd/libif_fir_ab_notan.so qo-100-tools_amd/libif_fir_dev.so --rounds 8 --reps 40 2>&1 | grep -v amdgpu.ids | tee $O/ab_tan_i16.txt
timeout -k 10 300 python3 tools/ab_inproc.py fir127_2p26 --variants 0 1001024 1002048 1003072 --rounds 16 --reps 60 2>&1 | grep -v amdgpu.ids | tee $O/ab_fir127_rows.txt
