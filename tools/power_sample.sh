#!/bin/bash
# power_sample.sh <tag> — board power, clocks and temperature sampled with rocm-smi while the headline step runs back to back
# for ~12 s, and while only its compute part runs (diagnostic launch without loads and stores): evidence for DESIGN §3.4 finding 8
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
export IF_FIR_DEBUG=1
sample() { # <seconds> <file>
  for i in $(seq 1 $(( $1 * 2 ))); do
    rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|Temperature \(Sensor (edge|junction|memory)" | tr '\n' ';' >> $2
    echo >> $2
    sleep 0.5
  done
}
echo "== idle" | tee $O/power.txt
sample 2 $O/idle.txt; tail -1 $O/idle.txt | cut -c1-400 | tee -a $O/power.txt
echo "== headline step back to back (bench.py --steps 20000)" | tee -a $O/power.txt
python3 bench.py --no-cpu-baseline --no-extra-configs --no-live-traffic --steps 20000 --warmup 10 > $O/bench_long.json 2>/dev/null &
BP=$!
sleep 6
sample 5 $O/load.txt
wait $BP
tail -3 $O/load.txt | cut -c1-400 | tee -a $O/power.txt
python3 -c "import json;d=json.loads(open('$O/bench_long.json').read().strip().splitlines()[-1]);print('ms/step %.4f frac %.4f' % (d['ms_per_step'], d['roofline']['frac']))" | tee -a $O/power.txt
echo "== compute only (diagnostic launch: loads and stores skipped, variant 1003), same instruction stream" | tee -a $O/power.txt
python3 - > $O/compute_only.txt 2>&1 <<'PY' &
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import __graft_entry__ as g
fir = g.load_pkg().if_fir
torch.cuda.set_device(0)
n = 1 << 28
x = torch.zeros(2 * n, dtype=torch.float32, device="cuda")
with fir.IfFir(fir.bpf_design(255), 4, 0, dev=True) as f:
    y = torch.empty(2 * f.out_count(n), dtype=torch.float32, device="cuda")
    f.set_tuning(1003)
    t0 = time.time()
    while time.time() - t0 < 12:
        ms = f.time_device(x.data_ptr(), y.data_ptr(), n, 0, 200)
    print("compute-only launch: %.4f ms" % ms)
PY
CP=$!
sleep 6
sample 4 $O/compute.txt
wait $CP
tail -3 $O/compute.txt | cut -c1-400 | tee -a $O/power.txt
cat $O/compute_only.txt | tail -1 | tee -a $O/power.txt
