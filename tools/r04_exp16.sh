#!/bin/bash
# r04_exp16.sh <tag> — round 4, batch 16 on ONE box (development tool): rocprofv3 --kernel-trace --stats of the filter-bank benchmark
# (tools/fbank_bench.py) for the all-slots launches: the profiler's average kernel durations beside the tool's event timings.
cd "$(dirname "$0")/.."
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp
for spec in "8 28 255 8" "16 28 255 8" "8 28 255 64 freq"; do
  tag=$(echo $spec | tr ' ' '_')
  rm -rf $O/kt_$tag
  timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$tag -- python3 $R/tools/fbank_bench.py $spec > $O/run_$tag.log 2>&1 || { echo "pass failed: $spec"; tail -3 $O/run_$tag.log; continue; }
  echo "== fbank_bench.py $spec: $(tail -1 $O/run_$tag.log | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("filter_bank_ms", d["filter_bank_ms"])')" | tee -a $O/bank_kernel_stats.txt
  f=$(find $O/kt_$tag -name "*kernel_stats.csv" | head -1)
  head -1 $f | tee -a $O/bank_kernel_stats.txt
  grep "fir_fft_kernel<4, true, false, \(true\|false\), \(8\|9\|17\)," $f | tee -a $O/bank_kernel_stats.txt
done
