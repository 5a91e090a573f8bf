#!/bin/bash
# r03_exp15.sh <tag> — all GPU tests with every multiple of 4 behind the decimate-by-4 tail, then a decimation sweep at 2^28 samples
cd "$(dirname "$0")/.."
O=gpurun_out/$1
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1
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
with fir.IfFir(fir.bpf_design(255), 1, 0, dev=True) as f0:
    f0.synth_device(x.data_ptr(), 0, n, 0)
    f0.synchronize()
xi = torch.empty(2 * n, dtype=torch.int16, device="cuda")
xi.copy_((x * 8000.0).round().clamp(-32768, 32767))
for t, fmt, nco in ((255, "f32", 0.0), (1023, "f32", 0.0), (255, "i16", 0.137), (255, "f32", 0.137)):
    taps = fir.bpf_design(t)
    for d in (1, 2, 3, 4, 6, 8, 12, 16, 20, 24, 32, 48, 64):
        with fir.IfFir(taps, d, 0, dev=True) as f:
            if fmt == "i16":
                f.set_input_format(fir.INPUT_I16)
            if nco:
                f.set_nco(nco)
            y = torch.empty(2 * f.out_count(n) + 16, dtype=torch.float32, device="cuda")
            src = xi if fmt == "i16" else x
            for _ in range(3):
                ms = f.time_device(src.data_ptr(), y.data_ptr(), n, 3, 20)
            byt = ((4 if fmt == "i16" else 8) + 8.0 / d) * n
            print("%4d taps %s nco=%s decimation %2d: %.4f ms  %.1f GS/s  %.0f GB/s (%.3f of 8 TB/s)" % (t, fmt, "on " if nco else "off", d, ms, n / ms / 1e6, byt / ms / 1e6, byt / ms / 1e6 / 8000))
PY
