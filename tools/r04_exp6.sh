#!/bin/bash
# r04_exp6.sh <tag> — round 4, batch 6 on ONE box (development tool): is the gain of configs[1] from more cached edge rows the
# memory-side cache serving the SAME 512 MB input again on the next launch?  The same A/B with the launches rotating over 6
# input/output buffer pairs (6 GB), next to the single-buffer form; and the 2-row kernel + tail phase the same way.
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
A=qo-100-tools_amd
L="$A/libif_fir_ab_full4.so $A/libif_fir_ab_full12.so $A/libif_fir_dev.so"
echo "== one buffer pair" | tee $O/ab_fir127_buffers.txt
timeout -k 10 400 python3 tools/ab_inproc.py fir127_2p26 --libs $L --variants 0 1003072 --rounds 8 --reps 60 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_fir127_buffers.txt
echo "== six buffer pairs in rotation" | tee -a $O/ab_fir127_buffers.txt
timeout -k 10 400 python3 tools/ab_inproc.py fir127_2p26 --libs $L --variants 0 1003072 --rounds 8 --reps 60 --buffers 6 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_fir127_buffers.txt
echo "== headline, three buffer pairs in rotation (7.5 GB)" | tee -a $O/ab_fir127_buffers.txt
timeout -k 10 400 python3 tools/ab_inproc.py fir255_dec4_2p28 --libs $A/libif_fir_ab_notan.so $A/libif_fir_dev.so --rounds 8 --reps 30 --buffers 3 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_fir127_buffers.txt
