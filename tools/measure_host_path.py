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
    f.reset()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); y = f.process(x); dt = time.perf_counter() - t0
        best = min(best, dt)
    print("if_fir_process host path: 2^26 samples, 255 taps /4: %.1f ms -> %.1f MSamples/s (%.1f GB/s over PCIe incl. pageable copies)"
          % (best * 1e3, n / best / 1e6, (8 * n + 2 * n) / best / 1e9))
