#!/usr/bin/env python3
"""mid_calls.py — time per call of mid-size calls (2^21 .. 2^27 samples, 255 taps, decimate by 4) with the block queue's tail phase
(short launches: a remainder of at most one block per SIMD runs one wave per SIMD) on and off (development variant 1000256 =
off).  Development tool; profiles/r03_queue_tail_ab.txt."""
import os
import sys
os.environ["IF_FIR_DEBUG"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import __graft_entry__ as g  # noqa: E402

fir = g.load_pkg().if_fir
torch.cuda.set_device(0)
nmax = 1 << 27
x = torch.empty(2 * nmax, dtype=torch.float32, device="cuda")
taps = fir.bpf_design(255)
with fir.IfFir(taps, 4, 0, dev=True) as f:
    f.synth_device(x.data_ptr(), 0, nmax, 0)
    f.synchronize()
    y = torch.empty(2 * f.out_count(nmax) + 16, dtype=torch.float32, device="cuda")
    for log2n in (21, 22, 23, 24, 25, 26, 27):
        n = 1 << log2n
        row = []
        for rnd in range(2):
            for var in (100, 1000256):
                f.set_tuning(var)
                f.reset()
                for _ in range(2):
                    ms = f.time_device(x.data_ptr(), y.data_ptr(), n, 50, 400 if log2n < 25 else 100)
                row.append(ms)
        blocks = (n + 3839) // 3840
        print("2^%d samples (%d blocks, %.2f per wave): tail on %.4f / %.4f ms   tail off %.4f / %.4f ms   (%+.1f %%)" %
              (log2n, blocks, blocks / 2048.0, row[0], row[2], row[1], row[3], 100.0 * ((row[0] + row[2]) / (row[1] + row[3]) - 1.0)))
