#!/usr/bin/env python3
"""stamps.py — per-wave start/end distribution of the persistent kernel (development tool)."""
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
taps = fir.bpf_design(taps_n)
torch.cuda.set_device(0)
x = torch.empty(2 * n, dtype=torch.float32, device="cuda")
with fir.IfFir(taps, decim, 0, dev=True) as f:
    y = torch.empty(2 * f.out_count(n), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    f.synth_device(x.data_ptr(), 0, n, 0)
    f.synchronize()
    f.debug_stamps()
    for v in variants:
        f.set_tuning(v)
        for _ in range(5):
            f.process_device(x.data_ptr(), y.data_ptr(), n)
        f.synchronize()
        ms = f.time_device(x.data_ptr(), y.data_ptr(), n, 2, 10)
        s = f.debug_stamps(2048).astype(np.int64)
        s = s[s[:, 1] > 0]
        t0 = s[:, 0].min()
        st = (s[:, 0] - t0) * 0.01   # us
        en = (s[:, 1] - t0) * 0.01
        dur = en - st
        clk = (s[:, 3] - s[:, 2]) / np.maximum(s[:, 1] - s[:, 0], 1) * 0.1   # GHz
        print("variant %d: %.4f ms (stamped build) waves=%d" % (v, ms, len(s)))
        print("  start us: min %.1f p50 %.1f p99 %.1f max %.1f" % (st.min(), np.median(st), np.percentile(st, 99), st.max()))
        print("  end   us: min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f" %
              (en.min(), np.percentile(en, 10), np.median(en), np.percentile(en, 90), en.max()))
        print("  dur   us: min %.1f p50 %.1f max %.1f ; mean/max-end = %.3f" % (dur.min(), np.median(dur), dur.max(), dur.mean() / en.max()))
        print("  clock GHz: min %.3f p50 %.3f max %.3f" % (clk.min(), np.median(clk), clk.max()))
        gw = np.arange(len(s))
        xcd = (gw // 4) % 8
        print("  mean end by block%8 group:", " ".join("%.0f" % en[xcd == k].mean() for k in range(8)))
        h, edges = np.histogram(en, bins=12)
        print("  end histogram:", " ".join("%d@%.0f" % (c, e) for c, e in zip(h, edges[:-1])))
