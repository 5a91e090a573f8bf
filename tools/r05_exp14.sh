#!/bin/bash
# r05_exp14.sh <tag> — round 5, batch 14 (development tool): per-phase stamps of the headline kernel with and without its loads / stores (where does pass 3,
# 44 % of a block's time for 23 % of its arithmetic, spend it?)
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
A=qo-100-tools_amd
export IF_FIR_DEBUG=1
for v in 0 1001 1002 1003; do
  timeout -k 10 200 python3 tools/fft_stamps.py fir255_dec4_2p28 $v $A/libif_fir_ab_stamps.so 2>&1 | grep -v amdgpu.ids | tee -a $O/stamps.txt
done
timeout -k 10 200 python3 tools/fft_stamps.py fir255_2p28 0 $A/libif_fir_ab_stamps.so 2>&1 | grep -v amdgpu.ids | tee -a $O/stamps.txt
