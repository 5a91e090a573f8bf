#!/bin/bash
# r04_exp14.sh <tag> — round 4, batch 14 on ONE box (development tool): SQ counters of the general (arbitrary-centre) bank forms at decimation
# 4 / 8 / 16 beside the all-slots form: instructions per block and the share of waiting.
cd "$(dirname "$0")/.."
R=$PWD
export TMPDIR=/tmp IF_FIR_DEBUG=1
O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp
for spec in "8 28 255 8 freq" "8 28 255 16 freq" "8 28 255 4 freq" "8 28 255 8"; do
  rm -rf $O/p
  timeout -k 5 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $O/p -- python3 $R/tools/fbank_bench.py $spec > $O/run.log 2>&1 || { echo "pass failed: $spec"; tail -3 $O/run.log; continue; }
  echo "== fbank_bench.py $spec" | tee -a $O/pmc_general.txt
  python3 $R/tools/pmc_summary.py $O/p | grep -A9 "fir_fft_kernel<4, true, false, \(true\|false\), \(5\|8\|9\|17\)," | tee -a $O/pmc_general.txt
done
