"""Generates tests/golden/if_fir_golden.npz — run in the dev container only (needs scipy).

PARITY UNPINNED: the reference (vankxr/qo-100-tools) has no FIR code, tests or vectors, so these
fixtures are produced by third-party tools (numpy + scipy.signal), NOT by the reference and NOT by this repo's
oracle: they pin the oracle (tests/test_oracle.py) and, through it, the HIP kernels.

Contents (all little-endian):
  taps_{127,255,1023}      float32   scipy.signal.firwin(T,[0.3,0.5],window='blackman',pass_zero=False,scale=True)
  x                        float32   4096 IQ samples, interleaved: SPEC §5 generator restated with numpy (channel 0)
  y_T{T}_D{D}              float64   interleaved I/Q:  scipy.signal.upfirdn(h, x, down=D)[:ceil(N/D)]
  ctaps_255, yc_T255_D{1,4}  complex taps: scipy low-pass (bandwidth 0.1) shifted to +0.2, outputs by upfirdn
  xr / yr_T127             float32/float64   real-sample case (BASELINE configs[0] shape, 4096 samples): lfilter
"""
import os

import numpy as np
import scipy.signal as ss

HERE = os.path.dirname(os.path.abspath(__file__))


def synth_numpy(n, channel=0, first=0):
    idx = np.arange(first, first + n, dtype=np.uint64)
    seed = np.uint64(0x5130303100000000 + channel)
    with np.errstate(over="ignore"):
        z = seed + (idx + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    ui = (((z >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)) - np.float32(0.5)) * np.float32(0.5)
    uq = ((((z >> np.uint64(16)) & np.uint64(0xFFFFFF)).astype(np.float32) * np.float32(2.0 ** -24))
          - np.float32(0.5)) * np.float32(0.5)
    i5 = np.arange(5, dtype=np.float64)
    ti = (0.5 * np.cos(2 * np.pi * 0.2 * i5) + 0.5 * np.cos(2 * np.pi * 0.4 * i5)).astype(np.float32)
    tq = (0.5 * np.sin(2 * np.pi * 0.2 * i5) + 0.5 * np.sin(2 * np.pi * 0.4 * i5)).astype(np.float32)
    p = (idx % np.uint64(5)).astype(np.int64)
    out = np.empty(2 * n, dtype=np.float32)
    out[0::2] = ti[p] + ui
    out[1::2] = tq[p] + uq
    return out


def main():
    n = 4096
    d = {}
    x = synth_numpy(n)
    d["x"] = x
    xc = x.view(np.complex64).astype(np.complex128)
    for t in (127, 255, 1023):
        h = ss.firwin(t, [0.3, 0.5], window="blackman", pass_zero=False, scale=True).astype(np.float32)
        d["taps_%d" % t] = h
        for dec in (1, 4):
            y = ss.upfirdn(h.astype(np.float64), xc, down=dec)[:(n + dec - 1) // dec]
            d["y_T%d_D%d" % (t, dec)] = np.ascontiguousarray(y).view(np.float64)
    # complex taps (channel selection): scipy low-pass prototype (two-sided bandwidth 0.1) shifted to +0.2 cycles/sample
    hl = ss.firwin(255, 0.1, window="blackman", scale=True)
    gc = (hl * np.exp(2j * np.pi * 0.2 * (np.arange(255) - 127))).astype(np.complex64)
    d["ctaps_255"] = gc.view(np.float32).copy()
    for dec in (1, 4):
        y = ss.upfirdn(gc.astype(np.complex128), xc, down=dec)[:(n + dec - 1) // dec]
        d["yc_T255_D%d" % dec] = np.ascontiguousarray(y).view(np.float64)
    # NCO (SPEC §3.2): integer-phase mix by -0.2 cycles/sample (band centre to 0) ahead of the scipy low-pass prototype
    pw = int(round(0.2 * 2 ** 32)) % 2 ** 32
    d["nco_phase_word"] = np.array([pw], dtype=np.uint64)
    a = np.arange(n, dtype=np.uint64)
    frac = ((a * np.uint64(pw)) & np.uint64(0xFFFFFFFF)).astype(np.float64) / 2.0 ** 32
    xm = xc * np.exp(-2j * np.pi * frac)
    d["taps_lp_255"] = hl.astype(np.float32)
    for dec in (1, 4):
        y = ss.upfirdn(d["taps_lp_255"].astype(np.float64), xm, down=dec)[:(n + dec - 1) // dec]
        d["ynco_T255_D%d" % dec] = np.ascontiguousarray(y).view(np.float64)
    xr = x[0::2].copy()
    d["xr"] = xr
    d["yr_T127"] = ss.lfilter(d["taps_127"].astype(np.float64), 1.0, xr.astype(np.float64))
    np.savez_compressed(os.path.join(HERE, "if_fir_golden.npz"), **d)
    print("wrote", os.path.join(HERE, "if_fir_golden.npz"), {k: v.shape for k, v in d.items()})


if __name__ == "__main__":
    main()
