#!/usr/bin/env python3
"""fft_clock.py [workload] [variants...] — in-kernel shader clock and per-wave run time of the overlap-save kernel
(development tool): every wave stamps s_memrealtime (100 MHz) and s_memtime (shader clock) at its start and end; the
clock is the quotient over the launch.  Variants as in tools/sweep.py (1000 + diagnostic bits)."""
import os
os.environ.setdefault("IF_FIR_DEBUG", "1")   # development tool: diagnostic tuning variants allowed
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import __graft_entry__ as g  # noqa: E402
import bench  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "fir255_dec4_2p28"
variants = [int(v) for v in sys.argv[2:]] or [0]
taps_n, decim, log2n, _ = bench.WORKLOADS[wl]
n = 1 << log2n
fir = g.load_pkg().if_fir
torch.cuda.set_device(0)
x = torch.empty(2 * n, dtype=torch.float32, device="cuda")
with fir.IfFir(fir.bpf_design(taps_n), decim, 0, dev=True) as f:
    f.set_backend(fir.BACKEND_HIP_FFT)
    y = torch.empty(2 * f.out_count(n), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    f.synth_device(x.data_ptr(), 0, n, 0)
    f.synchronize()
    f.debug_stamps()
    for v in variants:
        f.set_tuning(v)
        ms = [f.time_device(x.data_ptr(), y.data_ptr(), n, 0, 20) for _ in range(3)][-1]   # ~60 launches back to back
        st = f.debug_stamps(2048).astype(np.int64)
        ok = st[:, 1] > st[:, 0]
        st = st[ok]
        dur = (st[:, 1] - st[:, 0]) * 0.01                       # us
        clk = (st[:, 3] - st[:, 2]) / np.maximum(dur, 1e-9) / 1e3  # GHz
        span = (st[:, 1].max() - st[:, 0].min()) * 0.01
        print("%s variant %4d: %.4f ms/launch | waves %d | wave run time us min/median/max %.0f/%.0f/%.0f | launch span %.0f us | "
              "shader clock GHz min/median/max %.3f/%.3f/%.3f" %
              (wl, v, ms, ok.sum(), dur.min(), np.median(dur), dur.max(), span, clk.min(), np.median(clk), clk.max()), flush=True)
