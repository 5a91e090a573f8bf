#!/bin/bash
# r04_exp15.sh <tag> — round 4, batch 15 on ONE box (development tool): HBM traffic of the filter-bank launches from the TCC counters
# (FETCH_SIZE and WRITE_SIZE in their own rocprofv3 --pmc passes; read bytes = 2 x FETCH_SIZE KiB on gfx950, MI355X_MICROARCH.md) beside
# their algorithmic bytes.
cd "$(dirname "$0")/.."
R=$PWD
export TMPDIR=/tmp IF_FIR_DEBUG=1
O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp
for spec in "8 28 255 8" "16 28 255 8" "8 28 255 8 freq" "8 28 255 16 freq" "16 28 255 16" "8 28 255 4"; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/p
    timeout -k 5 150 rocprofv3 --pmc $ctr --output-format csv -d $O/p -- python3 $R/tools/fbank_bench.py $spec > $O/run.log 2>&1 || { echo "pass failed: $spec $ctr"; tail -3 $O/run.log; continue; }
    echo "== fbank_bench.py $spec ($ctr)" | tee -a $O/traffic_bank.txt
    python3 $R/tools/pmc_summary.py $O/p | grep -A1 "fir_fft_kernel<4, true, false, \(true\|false\), \(4\|5\|8\|9\|16\|17\)," | tee -a $O/traffic_bank.txt
  done
done
