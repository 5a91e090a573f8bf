#!/bin/bash
# r04_exp11.sh <tag> — round 4, batch 11 on ONE box (development tool): does the ORDER of bench.py's comparison measurements move the
# headline?  The driver's form of the command, alternating: extras after the timed steps (the default since round 4) / ahead of them
# (--extras-first, the order up to round 3), four times each; then the filter bank's counters (tools/pmc_bank.sh).
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
for k in 1 2 3 4; do
  for mode in "" "--extras-first"; do
    timeout -k 10 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 $mode > $O/b.json 2> $O/b.err || { echo "bench failed"; tail -3 $O/b.err; exit 1; }
    python3 - "$mode" $O/b.json <<'PY' | tee -a $O/bench_order.txt
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
r = d["roofline"]
fb = d["extra"].get("filter_bank", {})
print("order %-14s ms_per_step %.4f frac %.4f kernel median %.4f min %.4f cold %.4f | fir127 %.4f fir1023 %.4f dec3 %.4f bank8 %s" % (
    sys.argv[1] or "extras-after", d["ms_per_step"], r["frac"], r["kernel_ms_median"], r["kernel_ms_min"], r["cold_kernel_ms"],
    d["extra"]["configs"]["fir127_2p26"]["auto"]["frac"], d["extra"]["configs"]["fir1023_2p28"]["auto"]["frac"],
    d["extra"]["configs"]["fir255_dec3_2p28"]["auto"]["frac"], fb.get("kernel_ms")))
PY
  done
done
timeout -k 10 600 bash tools/pmc_bank.sh $1 2>&1 | tee $O/pmc_bank.txt | tail -40
