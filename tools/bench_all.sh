#!/bin/bash
# bench_all.sh <tag> — the round's result table: bench.py on every workload (settled rate, no CPU leg), one JSON line each
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
for w in fir255_dec4_2p28 fir127_2p26 fir255_2p28 fir1023_2p28 fir255_dec4_i16_2p28 fir255_dec4_nco_2p28 fir1023_dec8_2p28 fir255_dec2_2p28 fir2047_dec8_2p26 fir255_dec3_2p28 fir255_dec9_2p28 fir511_dec3_2p28; do
  python3 bench.py --workload $w --no-cpu-baseline --no-extra-configs --steps 50 --warmup 10 > $O/bench_$w.json 2> $O/bench_$w.err
  python3 - "$O/bench_$w.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("%-22s %-10s %.4f ms (median %.4f min %.4f)  %.1f GS/s  frac %.3f  whole_output %s" % (
    d["config"]["name"], d["config"]["backend"], r["kernel_ms"], r["kernel_ms_median"], r["kernel_ms_min"], d["value"] / 1e3,
    r["frac"], d["parity"]["whole_output"].get("ok")))
PY
done | tee $O/bench_all.txt
