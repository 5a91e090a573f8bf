#!/bin/bash
# r05_exp25.sh <tag> — round 5, batch 25 (development tool): the store offsets of the tails that keep every sub-th output in 32-bit arithmetic with 24-bit
# multiplies (libif_fir_dev.so) against the 64-bit form (libif_fir_ab_prev.so); the GPU tests that cover those tails first.
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
A=qo-100-tools_amd
export IF_FIR_DEBUG=1
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -2 | tee $O/pytest.txt
grep -q " passed" $O/pytest.txt || exit 1
LIBS="$A/libif_fir_ab_prev.so $A/libif_fir_dev.so"
for w in fir1023_dec8_2p28 fir2047_dec8_2p26 fir255_dec12_2p28 fir255_dec6_2p28 fir255_dec9_2p28 fir255_dec4_2p28; do
  timeout -k 10 300 python3 tools/ab_inproc.py $w --libs $LIBS --rounds 10 --reps 40 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
done
timeout -k 10 600 python3 tools/fbank_ab.py --libs $LIBS --cases 12:8:freq 20:8:freq 64:8:freq 2>&1 | grep -v amdgpu.ids | tee $O/fbank_ab.txt
