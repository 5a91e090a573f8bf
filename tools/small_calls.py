#!/usr/bin/env python3
"""small_calls.py [tuning] [decimation] [taps] — time per call of back-to-back small calls on one context (development tool): 2^16 ... 2^24 samples, 255 taps, decimate-by-4
by default; the call rate a chunked stream sees.  tuning (e.g. 1262144 = single-round launches off, IF_FIR_DEBUG=1) is applied to the context."""
import os
import sys
import time

os.environ.setdefault("IF_FIR_DEBUG", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import __graft_entry__ as g  # noqa: E402

tuning = int(sys.argv[1]) if len(sys.argv) > 1 else 0
decim = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ntaps = int(sys.argv[3]) if len(sys.argv) > 3 else 255
fir = g.load_pkg().if_fir
torch.cuda.set_device(0)
taps = fir.bpf_design(ntaps)
for log2n in (14, 16, 18, 20, 21, 22, 23, 24):
    n = 1 << log2n
    x = torch.empty(2 * n, dtype=torch.float32, device="cuda")
    with fir.IfFir(taps, decim, 0, dev=True) as f:
        if tuning:
            f.set_tuning(tuning)
        y = torch.empty(2 * f.out_count(n), dtype=torch.float32, device="cuda")
        f.synth_device(x.data_ptr(), 0, n, 0)
        f.synchronize()
        for _ in range(20):
            f.process_device(x.data_ptr(), y.data_ptr(), n)
        f.synchronize()
        reps = 2000 if log2n < 22 else 300
        t0 = time.perf_counter()
        for _ in range(reps):
            f.process_device(x.data_ptr(), y.data_ptr(), n)
        t_issue = time.perf_counter() - t0
        f.synchronize()
        t = time.perf_counter() - t0
        print("tuning %d D=%d T=%d n=2^%d: %.2f us per call (host issue %.2f us) -> %.1f GS/s" %
              (tuning, decim, ntaps, log2n, t / reps * 1e6, t_issue / reps * 1e6, n * reps / t / 1e9), flush=True)
