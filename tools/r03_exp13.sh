#!/bin/bash
# r03_exp13.sh <tag> — composite decimations behind the decimating tails (12, 20, ..., 60 = 4 x odd; 24, 40, 48, 56 = 8 x 3, 5, 6, 7):
# parity tests, then time per 2^28-sample pass against the selecting store (development variant 3000 where it applies, and the
# neighbouring decimations that have no tail)
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
python -m pytest tests -m gpu -x -q -k "any_decimation or chunked_equals or mc_threads or full_size_fft or random_config or spills" > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
export IF_FIR_DEBUG=1
python3 - <<'PY' | tee $O/times.txt
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import __graft_entry__ as g
fir = g.load_pkg().if_fir
torch.cuda.set_device(0)
n = 1 << 28
x = torch.empty(2 * n, dtype=torch.float32, device="cuda")
taps = fir.bpf_design(255)
with fir.IfFir(taps, 1, 0, dev=True) as f0:
    f0.synth_device(x.data_ptr(), 0, n, 0)
    f0.synchronize()
for d in (4, 12, 20, 28, 60, 8, 24, 40, 48, 56, 6, 10, 3, 5):
    with fir.IfFir(taps, d, 0, dev=True) as f:
        y = torch.empty(2 * f.out_count(n) + 16, dtype=torch.float32, device="cuda")
        row = []
        for var in (0, 3000):
            f.set_tuning(var)
            for _ in range(2):
                ms = f.time_device(x.data_ptr(), y.data_ptr(), n, 3, 20)
            row.append(ms)
        print("255 taps, decimation %2d, 2^28 samples: %.4f ms (variant 3000: %.4f ms)  %.1f GS/s" % (d, row[0], row[1], n / row[0] / 1e6))
PY
