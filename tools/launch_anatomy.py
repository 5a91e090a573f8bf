#!/usr/bin/env python3
"""launch_anatomy.py [workload ...] — where the fixed part of a launch goes (development tool; round 5): every wave stamps s_memrealtime
behind the table copy (its start) and at its end; with the launch period from HIP events this gives, per launch: the time outside every
wave's life (launch gap + dispatch + table copy), the spread of the waves' starts, and the idle wave-time at the end (mean distance of a
wave's end from the last end)."""
import os
os.environ.setdefault("IF_FIR_DEBUG", "1")
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import __graft_entry__ as g  # noqa: E402
import bench  # noqa: E402

fir = g.load_pkg().if_fir
torch.cuda.set_device(0)
for wl in (sys.argv[1:] or ["fir255_dec4_2p28", "fir127_2p26"]):
    taps_n, decim, log2n, _ = bench.WORKLOADS[wl]
    n = 1 << log2n
    x = torch.empty(2 * n, dtype=torch.float32, device="cuda")
    with fir.IfFir(fir.bpf_design(taps_n), decim, 0, dev=True) as f:
        f.set_backend(fir.BACKEND_HIP_FFT)
        y = torch.empty(2 * f.out_count(n), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        f.synth_device(x.data_ptr(), 0, n, 0)
        f.synchronize()
        f.debug_stamps()
        ms = [f.time_device(x.data_ptr(), y.data_ptr(), n, 0, 40) for _ in range(4)][-1]
        st = f.debug_stamps(2048).astype(np.int64)
        st = st[st[:, 1] > st[:, 0]]
        t0, t1 = st[:, 0] * 0.01, st[:, 1] * 0.01            # us
        span = t1.max() - t0.min()
        print("%s: period %.1f us | span first start .. last end %.1f us | outside = %.1f us | starts spread %.1f us (median %.1f after the first) | "
              "ends: median %.1f us, 10%% %.1f us, 90%% %.1f us before the last; idle wave-time at the end %.1f us = %.1f %% of the period | "
              "wave life min/median/max %.0f/%.0f/%.0f us" %
              (wl, ms * 1e3, span, ms * 1e3 - span, t0.max() - t0.min(), np.median(t0) - t0.min(), np.median(t1.max() - t1),
               np.percentile(t1.max() - t1, 10), np.percentile(t1.max() - t1, 90), np.mean(t1.max() - t1), 100 * np.mean(t1.max() - t1) / (ms * 1e3),
               (t1 - t0).min(), np.median(t1 - t0), (t1 - t0).max()), flush=True)
    del x, y
