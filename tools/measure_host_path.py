#!/usr/bin/env python3
"""PCIe-inclusive rate of if_fir_process (host buffers: H2D + kernel + D2H) — noted in DESIGN.md, never `value`."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as g
fir = g.load_pkg().if_fir
n = 1 << 26
taps = fir.bpf_design(255)
x = np.random.default_rng(1).standard_normal(2 * n).astype(np.float32)   # any input: this tool times copies + kernel
with fir.IfFir(taps, 4, n) as f:
    f.process(x[:2 * (1 << 20)])
    for kind in ("pageable", "page-locked (if_fir_host_alloc)"):
        f.reset()
        if kind == "pageable":
            xin, yout = x, np.empty(2 * f.out_count(n), dtype=np.float32)
        else:
            xin, yout = f.host_alloc(2 * n), f.host_alloc(2 * f.out_count(n))
            xin[:] = x
        best = 1e9
        for _ in range(4):
            f.reset()
            t0 = time.perf_counter(); f.process_into(xin, yout); dt = time.perf_counter() - t0
            best = min(best, dt)
        print("if_fir_process, %s host buffers: 2^26 samples, 255 taps /4: %.1f ms -> %.1f MSamples/s (%.1f GB/s over PCIe)"
              % (kind, best * 1e3, n / best / 1e6, (8 * n + 2 * n) / best / 1e9))
