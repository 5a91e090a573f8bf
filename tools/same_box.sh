#!/bin/bash
# same_box.sh <tag> — the memory floor and the filter on ONE box, one after the other (boxes differ by +-3 %): the
# microbenchmarks of tools/ubench_mem2.hip (copies, the 4:1 mix one-shot and as persistent 4096-point blocks), then
# bench.py sustained over 3000 steps and in its default form.
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
./tools/ubench_mem2 copy > $O/ubench_copy.txt 2>&1
./tools/ubench_mem2 blocks > $O/ubench_blocks.txt 2>&1
python3 bench.py --no-cpu-baseline --no-extra-configs --steps 3000 --warmup 200 > $O/bench_3000.json 2> /dev/null
python3 bench.py > $O/bench_default.json 2> /dev/null
{
echo "# tools/same_box.sh: everything below ran on one box, back to back"
grep "oneshot 1xf4\|mix 4:1\|hipMemcpy" $O/ubench_copy.txt
grep "pts=4096" $O/ubench_blocks.txt | grep "map=1 run=8\|map=0 run=1" | grep "LW= 8 SW= 8\|ntLS"
grep "ntLS\|ntS " $O/ubench_blocks.txt | grep "pts=4096"
python3 - "$O" <<'PY'
import json, sys
for f in ("bench_3000", "bench_default"):
    d = json.loads(open("%s/%s.json" % (sys.argv[1], f)).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("bench.py %-13s steps %4d: %.1f GS/s  %.4f ms/step  frac %.4f  (kernel median %.4f min %.4f ms)" %
          (f, d["steps"], d["value"] / 1e3, d["ms_per_step"], r["frac"], r["kernel_ms_median"], r["kernel_ms_min"]))
PY
} | tee $O/same_box.txt
