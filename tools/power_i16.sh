#!/bin/bash
# power_i16.sh <tag> — board power and shader clock (rocm-smi, every 0.5 s) while the headline step runs back to back with int16 input and,
# beside it, with float32 input: is the int16 instantiation at the package power limit like the float32 one?  (development tool; round 5)
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
export IF_FIR_DEBUG=1
smi() { rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Package Power|sclk" | sed -e 's/.*: //' | tr '\n' ' '; echo; }
for wl in fir255_dec4_i16_2p28 fir255_dec4_2p28; do
  python3 bench.py --workload $wl --no-cpu-baseline --no-extra-configs --no-live-traffic --steps 24000 --warmup 10 > $O/bench_$wl.json 2>/dev/null &
  BP=$!
  sleep 7
  rm -f $O/smi_$wl.txt
  for i in $(seq 1 8); do smi >> $O/smi_$wl.txt; sleep 0.5; done
  wait $BP
  python3 - $wl $O <<'PY'
import json, re, sys
wl, o = sys.argv[1:3]
d = json.loads(open("%s/bench_%s.json" % (o, wl)).read().strip().splitlines()[-1])
w, clk = [], []
for line in open("%s/smi_%s.txt" % (o, wl)):
    m = re.search(r"\((\d+)Mhz\).*?([\d.]+)\s*$", line)
    if m:
        clk.append(int(m.group(1))); w.append(float(m.group(2)))
print("%-24s %.4f ms/step  frac %.4f  package power %4.0f W (min %4.0f max %4.0f)  sclk %d-%d MHz" %
      (wl, d["ms_per_step"], d["roofline"]["frac"], sum(w) / max(1, len(w)), min(w or [0]), max(w or [0]), min(clk or [0]), max(clk or [0])))
PY
done
