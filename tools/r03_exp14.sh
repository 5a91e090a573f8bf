#!/bin/bash
# (historical: the one-channel filter-bank route for decimation 8 / 16 / 32 / 64 compared here was removed later in round 3 --
# every multiple of 4 now runs behind the decimate-by-4 tail; kept as the record of how the profiles/ file was produced)
# r03_exp14.sh <tag> — multiples of 8 / 16 through the decimate-by-4 tail keeping every 2nd, 4th, ... output (IF_FIR_EXP_TAIL4=1)
# against the one-channel filter-bank route they take now
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
export IF_FIR_DEBUG=1
for e in 0 1 0 1; do
if [ $e = 1 ]; then export IF_FIR_EXP_TAIL4=1; else unset IF_FIR_EXP_TAIL4; fi
python3 - $e <<'PY' | tee -a $O/times.txt
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
import __graft_entry__ as g
fir = g.load_pkg().if_fir
torch.cuda.set_device(0)
n = 1 << 28
x = torch.empty(2 * n, dtype=torch.float32, device="cuda")
with fir.IfFir(fir.bpf_design(255), 1, 0, dev=True) as f0:
    f0.synth_device(x.data_ptr(), 0, n, 0)
    f0.synchronize()
for t in (255, 1023):
    taps = fir.bpf_design(t)
    for d in (8, 16, 32, 64, 24, 48):
        with fir.IfFir(taps, d, 0, dev=True) as f:
            y = torch.empty(2 * f.out_count(n) + 16, dtype=torch.float32, device="cuda")
            for _ in range(3):
                ms = f.time_device(x.data_ptr(), y.data_ptr(), n, 3, 20)
            f.process_device(x.data_ptr(), y.data_ptr(), n); f.synchronize()
            ck = float(y[:2 * f.out_count(n)].double().abs().sum())
            print("tail4=%s  %4d taps, decimation %2d: %.4f ms  ck %.6e" % (sys.argv[1], t, d, ms, ck))
PY
done
