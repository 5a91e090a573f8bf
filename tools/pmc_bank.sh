#!/bin/bash
# pmc_bank.sh <tag> — SQ counters of the filter-bank kernels (8 channels at decimation 8 -- since round 4 the all-slots form, kernel 9 --, 16 channels at decimation 16) beside the
# single-channel decimate-by-4 kernel: where do the bank tails spend their cycles?  (development tool; counter sets in their
# own rocprofv3 --pmc passes, no trace domains)
cd "$(dirname "$0")/.."
R=$PWD
export TMPDIR=/tmp IF_FIR_DEBUG=1
O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp
for spec in "8 28 255 8" "16 28 255 16"; do
  set -- $spec
  for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" \
              "SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR"; do
    rm -rf $O/p
    timeout -k 5 150 rocprofv3 --pmc $ctrs --output-format csv -d $O/p -- python3 $R/tools/fbank_bench.py $spec > $O/run.log 2>&1 || { echo "pass failed: $ctrs"; tail -3 $O/run.log; continue; }
    echo "== bank $1 channels, decimation $4: $(tail -1 $O/run.log | cut -c1-10) ..."
    python3 $R/tools/pmc_summary.py $O/p | grep -A9 "fir_fft_kernel<4, true, false, false, \(8\|9\|16\),"
  done
done
