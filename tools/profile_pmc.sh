#!/bin/bash
# profile_pmc.sh <tag> — PMC tables for profiles/ (run on the GPU box through gpurun): SQ / LDS / TA / TCP counters of the
# overlap-save kernel (tuning variant 100) and of the direct-form kernel (variant 0) on the headline workload, each
# counter set in its own rocprofv3 --pmc pass (no trace domains next to --pmc).
cd "$(dirname "$0")/.."
R=$PWD
TAG=$1
O=$R/gpurun_out/$TAG
mkdir -p $O
{
echo "# rocprofv3 --pmc passes, tools/sweep.py fir255_dec4_2p28 <variant>; means over the full-size launches of each kernel"
echo "# SQ_* counters are in quad-cycles summed over all waves / SIMDs (MI355X_MICROARCH.md); TA_*/TCP_* summed over the CUs"
bash tools/pmc_variants.sh fir255_dec4_2p28 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU -- 100 0
bash tools/pmc_variants.sh fir255_dec4_2p28 SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -- 100 0
echo "# TA / TCP passes (one derived counter per pass: combined sets made the profiler abort on this pool)"
for c in TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum; do
  PMC_TIMEOUT=60 bash tools/pmc_variants.sh fir255_dec4_2p28 $c -- 100 || break
done
} > $O/pmc_table.txt 2>&1
grep -c mean $O/pmc_table.txt
