#!/bin/bash
# r03_exp7.sh <tag> — GPU tests, then the default bench line (with the live traffic passes) as the driver runs it
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu 2>&1 | tee $O/pytest.txt
grep -q " passed" $O/pytest.txt && ! grep -q "failed\|error" $O/pytest.txt || exit 1
( time timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err ) 2>&1 | grep real
python3 - "$O/bench_driver_form.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("value %.1f GS/s  ms/step %.4f (cold %.4f)  frac %.4f (cold %.4f)  traffic %.4g B = %.4f x algorithmic" %
      (d["value"] / 1e3, d["ms_per_step"], d["cold_ms_per_step"], r["frac"], r["cold_frac"], r["traffic"], r["traffic_over_algorithmic"]))
print(r["traffic_source"][:260])
for k, v in d["extra"]["configs"].items():
    if isinstance(v, dict) and "auto" in v:
        print(k, v["auto"]["kernel_ms"], v["auto"]["frac"], v["auto"].get("traffic_over_algorithmic"))
PY
