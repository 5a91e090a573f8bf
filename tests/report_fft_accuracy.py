#!/usr/bin/env python3
"""Error of the overlap-save backend (and the direct form) against the float64 oracle on 2^20 samples (dev tool)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as g
fir = g.load_pkg().if_fir
oracle = g.load_oracle()
n = 1 << 20
rng = np.random.default_rng(0)
inputs = {"synthetic": oracle.synth_iq(n), "gaussian": rng.standard_normal(2 * n).astype(np.float32),
          "inband tone": np.exp(2j * np.pi * 0.2 * np.arange(n)).astype(np.complex64).view(np.float32)}
for t, d in [(255, 4), (255, 1), (1023, 1), (127, 1), (127, 4)]:
    taps = fir.bpf_design(t)
    for name, x in inputs.items():
        ref = oracle.fir_f64(taps, x, d)
        with fir.IfFir(taps, d, n) as f:
            f.set_backend(fir.BACKEND_HIP_FFT)
            yf = f.process(x)
            f.reset()
            f.set_backend(fir.BACKEND_HIP_DIRECT if t in (127, 255) else fir.BACKEND_HIP_GENERIC)
            yd = f.process(x)
        ef, ed = oracle.err_metrics(yf, ref), oracle.err_metrics(yd, ref)
        print("T=%4d D=%d %-12s fft l2=%.2e max=%.2e | direct l2=%.2e max=%.2e" % (t, d, name, ef[0], ef[1], ed[0], ed[1]), flush=True)
