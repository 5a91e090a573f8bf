#!/bin/bash
# energy_attr.sh <tag> — energy per packed-FP32 instruction class and per "16-point transform + 15 twiddles" in its two forms
# (tools/ubench_energy.hip): rocm-smi sampled while each variant runs alone at the overlap-save kernel's occupancy.
# energy per wave-group = (W - idle W) / (wave-groups per second).  Development tool (DESIGN §3.4 finding 12).
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
smi() { rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Package Power|sclk" | sed -e 's/.*: //' | tr '\n' ' '; echo; }
sample() { for i in $(seq 1 $1); do smi >> $2; sleep 0.5; done; }
: > $O/energy_attr.txt
./tools/ubench_energy check | tee -a $O/energy_attr.txt || exit 1
rm -f $O/s.txt; sample 5 $O/s.txt; cp $O/s.txt $O/idle.txt
for v in scale add mul fma old new old new; do
  ./tools/ubench_energy $v 8 > $O/run.txt 2>&1 & P=$!
  sleep 3; rm -f $O/s.txt; sample 8 $O/s.txt; wait $P
  python3 - $v $O/s.txt "$(tail -1 $O/run.txt)" $O/idle.txt <<'PY' | tee -a $O/energy_attr.txt
import re, sys
v, path, res, idle = sys.argv[1:5]
def rd(p):
    w, clk = [], []
    for line in open(p):
        m = re.search(r"\((\d+)Mhz\).*?([\d.]+)\s*$", line)
        if m:
            clk.append(int(m.group(1))); w.append(float(m.group(2)))
    return w, clk
w, clk = rd(path); wi, _ = rd(idle)
w, clk = w[1:], clk[1:]
W = sum(w) / max(1, len(w)); WI = sum(wi[1:]) / max(1, len(wi) - 1)
m = re.search(r"([\d.]+) G wave-groups/s", res)
g = float(m.group(1)) * 1e9 if m else 0
print("%-6s %6.0f W (idle %.0f) sclk %4d-%4d MHz | %s | %.2f nJ per wave-group above idle" % (v, W, WI, min(clk or [0]), max(clk or [0]), res.strip(), (W - WI) / g * 1e9 if g else 0))
PY
done
