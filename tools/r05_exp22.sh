#!/bin/bash
# r05_exp22.sh <tag> — round 5, batch 22 (development tool): how many waves of a workgroup should request their first block AHEAD of the table copy (the others
# request it behind the workgroup barrier)?  Tuning 1000000 + 16384 + (P << 16): P = 0, 2, 4, 6; 0 = all eight (the library so far).
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
export IF_FIR_DEBUG=1
V="0 1016384 1147456 1278528 1409600"
for w in fir127_2p26 fir255_dec4_2p24 fir255_dec4_2p28 fir255_2p28; do
  timeout -k 10 400 python3 tools/ab_inproc.py $w --variants $V --rounds 10 --reps 40 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
done
