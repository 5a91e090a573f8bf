#!/bin/bash
# r05_exp9.sh <tag> — round 5, batch 9 on ONE box (development tool): where the FULL-RATE pipeline spends a block (255 taps D = 1 runs at 0.69 of
# 8 TB/s, the same traffic as a copy that runs at 0.83): per-phase stamps of the wave loop (libif_fir_ab_stamps.so), the SQ counters of the
# full-rate kernel, and the launch without its loads / stores / both (tuning 1001 / 1002 / 1003).
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
A=qo-100-tools_amd
export IF_FIR_DEBUG=1
for w in fir255_2p28 fir255_dec4_2p28 fir1023_2p28; do
  timeout -k 10 200 python3 tools/fft_stamps.py $w 0 $A/libif_fir_ab_stamps.so 2>&1 | grep -v amdgpu.ids | tee -a $O/stamps.txt
done
for v in 100 1001 1002 1003; do
  timeout -k 10 200 python3 tools/sweep.py fir255_2p28 $v 2>&1 | grep variant | tee -a $O/diag.txt
done
{
bash tools/pmc_variants.sh fir255_2p28 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU -- 100
bash tools/pmc_variants.sh fir255_2p28 SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -- 100
} > $O/pmc_full_rate.txt 2>&1
grep -c mean $O/pmc_full_rate.txt
