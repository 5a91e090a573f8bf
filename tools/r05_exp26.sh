#!/bin/bash
# r05_exp26.sh <tag> — round 5, batch 26 (development tool): the selecting store's index with 24-bit multiplies (libif_fir_dev.so) against the 32-bit ones
# (libif_fir_ab_prev.so): decimation 5, 7, 25, and 1023 taps /3 (more than 767 taps: selecting store).
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
A=qo-100-tools_amd
export IF_FIR_DEBUG=1
LIBS="$A/libif_fir_ab_prev.so $A/libif_fir_dev.so"
timeout -k 10 300 python3 -m pytest tests -x -q -m gpu -k "any_decimation or random_configurations" 2>&1 | tail -2 | tee $O/pytest.txt
for w in fir255_dec5_2p28 fir255_dec7_2p28 fir255_dec25_2p28 fir1023_dec3_2p28; do
  timeout -k 10 300 python3 tools/ab_inproc.py $w --libs $LIBS --rounds 10 --reps 40 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
done
