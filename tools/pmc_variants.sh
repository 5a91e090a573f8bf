#!/bin/bash
# pmc_variants.sh <workload> <counters...> -- <variants...> : one rocprofv3 --pmc pass per tuning variant (development tool)
cd "$(dirname "$0")/.."
R=$PWD
export TMPDIR=/tmp
WL=$1; shift
CTRS=()
while [ "$1" != "--" ]; do CTRS+=("$1"); shift; done
shift
cd /tmp
for v in "$@"; do
  rm -rf $R/gpurun_out/pmcv_$v
  # a counter set the profiler rejects can leave it hanging after its abort: bound every pass
  timeout -k 5 ${PMC_TIMEOUT:-120} rocprofv3 --pmc "${CTRS[@]}" --output-format csv -d $R/gpurun_out/pmcv_$v -- python3 $R/tools/sweep.py $WL $v > $R/gpurun_out/pmcv_$v.log 2>&1 || { echo "== variant $v: pass failed or timed out (${CTRS[*]})"; exit 1; }
  echo "== variant $v: $(grep variant $R/gpurun_out/pmcv_$v.log | cut -c1-60)"
  python3 $R/tools/pmc_summary.py $R/gpurun_out/pmcv_$v | grep -A20 -E 'fir_fft|fir_direct'
done
