#!/usr/bin/env python3
"""direct_clock.py [workload] [seconds] — evidence for the direct-form kernel's ceiling: after `seconds` (default 2) of
back-to-back launches on the synthetic (random) stream, the in-kernel shader clock of every wave
(delta s_memtime / delta s_memrealtime x 100 MHz over the wave's whole run) and the FP32 rate that goes with it.
Prints a table for profiles/ (per-wave MHz min / p10 / median / p90 / max)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import __graft_entry__ as g  # noqa: E402
import bench  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "fir255_dec4_2p28"
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
taps_n, decim, log2n, _ = bench.WORKLOADS[wl]
n = 1 << log2n
fir = g.load_pkg().if_fir
torch.cuda.set_device(0)
x = torch.empty(2 * n, dtype=torch.float32, device="cuda")
with fir.IfFir(fir.bpf_design(taps_n), decim, 0, dev=True) as f:
    y = torch.empty(2 * f.out_count(n), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    f.synth_device(x.data_ptr(), 0, n, 0)
    f.synchronize()
    for name, backend in (("direct (fir_direct_wave_kernel / fir_direct_kernel)", fir.BACKEND_HIP_DIRECT),
                          ("overlap-save (fir_fft_kernel)", fir.BACKEND_HIP_FFT)):
        f.set_backend(backend)
        f.reset()
        f.debug_stamps()
        t0, launches, ms = time.time(), 0, 0.0
        while time.time() - t0 < seconds:
            ms = f.time_device(x.data_ptr(), y.data_ptr(), n, 0, 50)
            launches += 50
        st = f.debug_stamps(8192).astype(np.int64)
        st = st[st[:, 1] > st[:, 0]]
        if not len(st):
            print("%s | %s | %.4f ms/launch | this kernel writes no wave stamps (only the persistent kernels do)" % (wl, name, ms))
            continue
        dur = (st[:, 1] - st[:, 0]) * 0.01
        mhz = (st[:, 3] - st[:, 2]) / np.maximum(dur, 1e-9)
        flops = bench.algorithmic_flops_per_sample(taps_n, decim) * n
        print("%s | %s | %d launches back to back (%.1f s) | last 50: %.4f ms/launch" % (wl, name, launches, time.time() - t0, ms))
        print("  waves stamped %d | wave run time us min/median/max %.0f/%.0f/%.0f" % (len(st), dur.min(), np.median(dur), dur.max()))
        print("  in-kernel shader clock MHz: min %.0f  p10 %.0f  median %.0f  p90 %.0f  max %.0f" %
              (mhz.min(), np.percentile(mhz, 10), np.median(mhz), np.percentile(mhz, 90), mhz.max()))
        if backend == fir.BACKEND_HIP_DIRECT:
            tf = flops / (ms * 1e-3) / 1e12
            ceil_tf = 256 * 256 * np.median(mhz) * 1e6 / 1e12     # 256 CUs x 256 flop/clk/CU (4 SIMD32 x packed FMA)
            print("  executed FP32 rate %.1f TFLOP/s = %.1f %% of the 157.3 spec peak (2.4 GHz) = %.1f %% of the peak at the "
                  "measured median clock (%.1f TFLOP/s)" % (tf, tf / 1.573, 100 * tf / ceil_tf, ceil_tf))
