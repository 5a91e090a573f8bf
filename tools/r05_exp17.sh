#!/bin/bash
# r05_exp17.sh <tag> — round 5, batch 17 (development tool): the measurements behind two "priced, not built" items:
#  (a) decimation 5, 7, 25 through the selecting store beside decimation 3, 9 on the odd-decimation kernel and the full-rate pipeline itself;
#  (b) the decimation-8 bank's all-slots form at ONE wave per SIMD (tuning 1064: waves 4-7 of every workgroup leave at once; results stay correct) against two
#      -- what a form that keeps both parities' 128 values per lane in registers would have to live with.
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
export IF_FIR_DEBUG=1
for w in fir255_2p28 fir255_dec3_2p28 fir255_dec9_2p28 fir255_dec5_2p28 fir255_dec7_2p28 fir255_dec25_2p28; do
  timeout -k 10 300 python3 tools/ab_inproc.py $w --rounds 8 --reps 40 2>&1 | grep -v amdgpu.ids | tee -a $O/odd.txt
done
timeout -k 10 300 python3 tools/ab_inproc.py fir255_dec4_2p28 --variants 0 1064 --rounds 8 --reps 40 2>&1 | grep -v amdgpu.ids | tee -a $O/one_wave.txt
for spec in "8 28 255 8" "8 28 255 8 tuning=1064" "16 28 255 8" "16 28 255 8 tuning=1064" "16 28 255 16" "16 28 255 16 tuning=1064"; do
  timeout -k 10 300 python3 tools/fbank_bench.py $spec 2>&1 | tail -1 | cut -c1-600 | tee -a $O/one_wave.txt
done
