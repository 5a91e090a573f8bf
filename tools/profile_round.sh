#!/bin/bash
# profile_round.sh <tag> [workload] — the three rocprofv3 passes behind profiles/<tag>_* (run on the GPU box through gpurun):
# kernel trace + stats of the default bench command, then FETCH_SIZE and WRITE_SIZE in their own passes (no trace domains
# next to --pmc).  tools/collect_profiles.py condenses the result into profiles/.
set -e
TAG=$1
WL=${2:-fir255_dec4_2p28}
cd "$(dirname "$0")/.."
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --workload $WL > $O/bench.json 2> $O/bench.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pf -- python3 $R/bench.py --workload $WL --no-cpu-baseline --steps 5 --warmup 2 > $O/pf.json 2> $O/pf.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pw -- python3 $R/bench.py --workload $WL --no-cpu-baseline --steps 5 --warmup 2 > $O/pw.json 2> $O/pw.err
tail -1 $O/bench.json | cut -c1-300
