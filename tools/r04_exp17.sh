#!/bin/bash
# r04_exp17.sh <tag> — round 4, batch 17 on ONE box (development tool): HBM read traffic of the 16-channel decimation-8 call as ONE launch over
# virtual blocks against two launches (development launch 8192): does L2 serve the second read of every block?
cd "$(dirname "$0")/.."
R=$PWD
export TMPDIR=/tmp IF_FIR_DEBUG=1
O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp
for spec in "16 28 255 8" "16 28 255 8 tuning=1008192"; do
  rm -rf $O/p
  timeout -k 5 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/p -- python3 $R/tools/fbank_bench.py $spec > $O/run.log 2>&1 || { echo "pass failed: $spec"; tail -3 $O/run.log; continue; }
  echo "== fbank_bench.py $spec (FETCH_SIZE, KiB per launch; read bytes = 2 x)" | tee -a $O/traffic_both.txt
  python3 $R/tools/pmc_summary.py $O/p | grep -A1 "fir_fft_kernel<4, true, false, false, 9," | tee -a $O/traffic_both.txt
done
