#!/usr/bin/env python3
"""policy_sweep.py — times every backend that accepts a (taps, decimation) pair on 2^log2n resident samples, to place
the AUTO thresholds of if_fir_shim.cpp (development tool).  usage: python tools/policy_sweep.py [log2n=27]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import __graft_entry__ as g  # noqa: E402


def main():
    log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 27
    n = 1 << log2n
    fir = g.load_pkg().if_fir
    torch.cuda.set_device(0)
    x = torch.empty(2 * n, dtype=torch.float32, device="cuda")
    names = {fir.BACKEND_HIP_DIRECT: "direct", fir.BACKEND_HIP_TAPSPLIT: "tapsplit", fir.BACKEND_HIP_GENERIC: "generic",
             fir.BACKEND_HIP_FFT: "fft"}
    first = True
    for t, d in [(3, 1), (7, 1), (15, 1), (31, 1), (33, 1), (63, 1), (15, 4), (31, 4), (33, 4), (63, 4), (63, 2), (31, 2),
                 (63, 8), (127, 8), (129, 8), (255, 8), (127, 16), (255, 16), (257, 16), (511, 16), (511, 32), (1023, 32),
                 (1023, 64), (255, 64), (2047, 1), (3073, 1), (3075, 1), (4095, 1), (4095, 4), (4095, 16)]:
        taps = (np.random.default_rng(t).standard_normal(t) / np.sqrt(t)).astype(np.float32)
        with fir.IfFir(taps, d, 0, dev=True) as f:
            if first:
                f.synth_device(x.data_ptr(), 0, n, 0)
                f.synchronize()
                first = False
            y = torch.empty(2 * f.out_count(n) + 8, dtype=torch.float32, device="cuda")
            auto = names[f.get_backend()]
            res = {}
            for b in (fir.BACKEND_HIP_FFT, fir.BACKEND_HIP_TAPSPLIT, fir.BACKEND_HIP_DIRECT, fir.BACKEND_HIP_GENERIC):
                try:
                    f.set_backend(b)
                except fir.IfFirError:
                    continue
                if b == fir.BACKEND_HIP_GENERIC and t * n / d > 3e11:
                    continue
                if b == fir.BACKEND_HIP_TAPSPLIT and t * n / d > 2.5e11:   # > 30 ms per launch: 2 launches tell enough
                    f.reset()
                    res[names[b]] = f.time_device(x.data_ptr(), y.data_ptr(), n, 0, 2)
                    continue
                f.reset()
                res[names[b]] = f.time_device(x.data_ptr(), y.data_ptr(), n, 2, 5)
            best = min(res, key=res.get)
            floor = (8.0 + 8.0 / d) * n / 5.5e12 * 1e3
            print("T=%4d D=%2d  auto=%-8s best=%-8s  %s   (copy-rate floor %.3f ms)%s" %
                  (t, d, auto, best, "  ".join("%s %.3f" % (k, v) for k, v in res.items()), floor,
                   "" if auto == best or res[auto] <= 1.1 * res[best] else "   <-- AUTO is %.1fx slower" % (res[auto] / res[best])),
                  flush=True)


if __name__ == "__main__":
    main()
