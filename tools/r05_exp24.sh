#!/bin/bash
# r05_exp24.sh <tag> — round 5, batch 24 (development tool): per-phase stamps of the int16-input headline kernel (0.55 of its 6 B/sample: neither the VALU nor the
# memory is saturated) with and without its loads / stores, beside the float32 one; and of the kernel that keeps every second output of the decimate-by-4 tail (1023 taps /8).
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
A=qo-100-tools_amd
export IF_FIR_DEBUG=1
for v in 0 1001 1002; do
  timeout -k 10 200 python3 tools/fft_stamps.py fir255_dec4_i16_2p28 $v $A/libif_fir_ab_stamps.so 2>&1 | grep -v amdgpu.ids | tee -a $O/stamps.txt
done
timeout -k 10 200 python3 tools/fft_stamps.py fir255_dec4_2p28 0 $A/libif_fir_ab_stamps.so 2>&1 | grep -v amdgpu.ids | tee -a $O/stamps.txt
timeout -k 10 200 python3 tools/fft_stamps.py fir1023_dec8_2p28 0 $A/libif_fir_ab_stamps.so 2>&1 | grep -v amdgpu.ids | tee -a $O/stamps.txt
