#!/usr/bin/env python3
"""fft_stamps.py — phase timing of the overlap-save kernel's wave loop (development tool)."""
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
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 0   # 1001/1002/1003: skip loads / stores / both
lib_path = sys.argv[3] if len(sys.argv) > 3 else None    # a build with -DIF_FIR_FFT_STAMPS (tools/build_ab.sh stamps -DIF_FIR_FFT_STAMPS)
taps_n, decim, log2n, _ = bench.WORKLOADS[wl]
n = 1 << log2n
fir = g.load_pkg().if_fir
torch.cuda.set_device(0)
x = torch.empty(2 * n, dtype=torch.float32, device="cuda")
with fir.IfFir(fir.bpf_design(taps_n), decim, 0, dev=True, lib_path=lib_path) as f:
    f.set_backend(fir.BACKEND_HIP_FFT)
    f.set_tuning(variant)
    y = torch.empty(2 * f.out_count(n), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    f.synth_device(x.data_ptr(), 0, n, 0)
    f.synchronize()
    src = x
    if "_i16_" in wl:   # int16 input front-end: the same stream as int16 pairs
        src = (x * 32767.0).round().clamp(-32768, 32767).to(torch.int16)
        f.set_input_format(fir.INPUT_I16)
    if "_nco_" in wl:
        f.set_nco(1638.0 / 8192.0)
    f.debug_stamps()
    for _ in range(3):
        f.process_device(src.data_ptr(), y.data_ptr(), n)
    f.synchronize()
    raw = f.debug_stamps(2048).astype(np.int64).reshape(-1)
    s = raw[:8 * 32 * 8].reshape(8, 32, 8)
    c = raw[4096:4096 + 8 * 32 * 8].reshape(8, 32, 8)   # s_memtime (shader clock) beside s_memrealtime (100 MHz)
    for w in range(2):
        ok = s[w][:, 7] > 0
        if ok.sum() > 4:
            i0, i1 = np.flatnonzero(ok)[1], np.flatnonzero(ok)[-1]
            print("wave %d shader clock over iterations %d..%d: %.3f GHz" %
                  (w, i0, i1, (c[w][i1, 7] - c[w][i0, 0]) / ((s[w][i1, 7] - s[w][i0, 0]) * 10.0)))
    names = (["top->loaded", "pass1", "exch1+pass2", "exch2+take", "pass3+H+inv3", "exch2+inv2+exch1", "inv1+stores+loads"] if decim == 1 else
             ["top->loaded", "pass1", "exch1+pass2", "exch2", "pass3(+fold,issue)", "inverse", "stores"])
    print(wl, 'variant', variant)
    for w in range(4):
        d = np.diff(s[w], axis=1) * 0.01   # us
        tot = (s[w][:, 7] - s[w][:, 0]) * 0.01
        ok = s[w][:, 7] > 0
        if not ok.any():
            continue
        print("wave", w, "iterations", ok.sum(), "mean us per phase:",
              " ".join("%s=%.2f" % (nm, v) for nm, v in zip(names, d[ok].mean(axis=0))), "total=%.2f" % tot[ok].mean())
        gaps = (s[w][1:, 0] - s[w][:-1, 7]) * 0.01
        print("   loop-back gap us: %.2f ; per-iteration totals: %s" %
              (gaps[ok[1:]].mean(), " ".join("%.1f" % v for v in tot[ok][:12])))
