#!/bin/bash
# pmc_bank_freq.sh <tag> — SQ counters of the filter bank's general forms (8 channels at arbitrary centres, decimation 16 / 8 / 64: kernels
# <.., 17> and <.., true, 8>), counter sets in their own rocprofv3 --pmc passes, no trace domains (development tool; round 5)
cd "$(dirname "$0")/.."
R=$PWD
export TMPDIR=/tmp IF_FIR_DEBUG=1
O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp
for spec in "8 28 255 16 freq" "8 28 255 8 freq" "8 28 255 64 freq"; do
  set -- $spec
  for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" \
              "SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR"; do
    rm -rf $O/p
    timeout -k 5 200 rocprofv3 --pmc $ctrs --output-format csv -d $O/p -- python3 $R/tools/fbank_bench.py $spec > $O/run.log 2>&1 || { echo "pass failed: $ctrs"; tail -3 $O/run.log; continue; }
    echo "== bank $1 channels at arbitrary centres, decimation $4"
    python3 $R/tools/pmc_summary.py $O/p | grep -A9 "fir_fft_kernel<4, true, false, \(false, 17\|true, 8\),"
  done
done
