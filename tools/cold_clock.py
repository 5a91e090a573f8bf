#!/usr/bin/env python3
"""cold_clock.py — what "cold" means for the headline launch (development tool): in-kernel shader clock and launch time of the
k-th launch after the device has been idle for a second, from the wave stamps of if_fir_debug_stamps (s_memrealtime /
s_memtime at the start and end of every wave).  Beside it rocm-smi's socket power is too slow to resolve (0.5 s)."""
import os
os.environ.setdefault("IF_FIR_DEBUG", "1")
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import __graft_entry__ as g  # noqa: E402

n = 1 << 28
fir = g.load_pkg().if_fir
torch.cuda.set_device(0)
x = torch.empty(2 * n, dtype=torch.float32, device="cuda")
with fir.IfFir(fir.bpf_design(255), 4, 0, dev=True) as f:
    y = torch.empty(2 * f.out_count(n), dtype=torch.float32, device="cuda")
    f.synth_device(x.data_ptr(), 0, n, 0)
    f.synchronize()
    f.debug_stamps()

    def one(label):
        f.process_device(x.data_ptr(), y.data_ptr(), n)
        f.synchronize()
        st = f.debug_stamps(2048).astype(np.int64)
        st = st[st[:, 1] > st[:, 0]]
        dur = (st[:, 1] - st[:, 0]) * 0.01
        clk = (st[:, 3] - st[:, 2]) / np.maximum(dur, 1e-9) / 1e3
        span = (st[:, 1].max() - st[:, 0].min()) * 0.01
        print("%-34s launch span %6.1f us   shader clock GHz min/median/max %.3f/%.3f/%.3f" % (label, span, clk.min(), np.median(clk), clk.max()), flush=True)

    for rnd in range(2):
        time.sleep(1.0)
        for k in (1, 2, 3, 5, 10):
            one("idle 1 s, launch %d (sync each)" % k)
        # back to back without host synchronisation in between
        for burst in (25, 100, 400, 1000):
            for _ in range(burst):
                f.process_device(x.data_ptr(), y.data_ptr(), n)
            one("after %d more back-to-back launches" % burst)

# launch-by-launch times of 300 back-to-back launches after one second of idle (HIP events on the context's stream, no host
# synchronisation in between)
with fir.IfFir(fir.bpf_design(255), 4, 0) as f2:
    stream = torch.cuda.Stream()
    f2.set_stream(stream.cuda_stream)
    for rnd in range(2):
        torch.cuda.synchronize()
        time.sleep(1.0)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(301)]
        evs[0].record(stream)
        for k in range(300):
            f2.process_device(x.data_ptr(), y.data_ptr(), n)
            evs[k + 1].record(stream)
        torch.cuda.synchronize()
        ms = [evs[k].elapsed_time(evs[k + 1]) for k in range(300)]
        print("after 1 s idle, ms of launch 1..300 (event to event), averages over groups of 10:")
        print("  " + " ".join("%.3f" % (sum(ms[i:i + 10]) / 10) for i in range(0, 300, 10)))
        print("  first ten: " + " ".join("%.3f" % v for v in ms[:10]))
