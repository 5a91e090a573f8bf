#!/bin/bash
# profile_extras.sh <tag> — rocprofv3 kernel stats of the filter bank and the WB detector, and the memory / VALU
# microbenchmarks behind DESIGN.md §3.4/§3.6 (run on the GPU box through gpurun; copy the summaries into profiles/)
set -e
TAG=$1
cd "$(dirname "$0")/.."
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/${TAG}_extras
mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/fbank -- python3 $R/tools/fbank_bench.py 8 28 255 > $O/fbank.json 2> $O/fbank.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/wb -- python3 $R/tests/bench_wb_detect.py 262144 918 > $O/wb.json 2> $O/wb.err
$R/tools/ubench_mem > $O/ubench_mem.txt 2>&1
$R/tools/ubench_valu > $O/ubench_valu.txt 2>&1 || true
python3 $R/tools/small_calls.py > $O/small_calls.txt 2>&1
tail -1 $O/fbank.json | cut -c1-200
tail -1 $O/wb.json | cut -c1-200
