"""ctypes wrapper around oracle/libif_fir_oracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
PARITY UNPINNED: the reference (vankxr/qo-100-tools) has no implementation of this path
(SURVEY.md §0/§8c); the oracle follows docs/SPEC.md and is cross-checked against numpy/scipy.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_f32p = ctypes.POINTER(ctypes.c_float)
_f64p = ctypes.POINTER(ctypes.c_double)


def build(arch="x86-64-v3", out=None):
    """Compile the oracle with gcc. Returns the path of the shared object."""
    out = out or os.path.join(_HERE, "libif_fir_oracle.so")
    src = os.path.join(_HERE, "if_fir_oracle.c")
    if os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(src):
        return out
    cmd = ["gcc", "-O3", "-march=" + arch, "-fopenmp", "-fPIC", "-std=gnu11", "-shared", "-o", out, src, "-lm"]
    subprocess.check_call(cmd)
    return out


def _bind(lib):
    u8, u32, u64, i32 = ctypes.c_uint8, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int32
    lib.oracle_bpf_design.argtypes = [_f32p, u32, ctypes.c_double, ctypes.c_double, u32]
    lib.oracle_bpf_design.restype = ctypes.c_int
    lib.oracle_synth_tone_table.argtypes = [_f32p]
    lib.oracle_synth_iq.argtypes = [_f32p, u64, u64, u32]
    lib.oracle_out_count.argtypes = [u64, u64, u32]
    lib.oracle_out_count.restype = u64
    lib.oracle_update_history.argtypes = [_f32p, _f32p, u32, _f32p, u64]
    lib.oracle_fir_c64_f64.argtypes = [_f32p, u32, u32, _f32p, u64, _f32p, u64, _f64p]
    lib.oracle_fir_c64_f64.restype = u64
    lib.oracle_fir_r32_f64.argtypes = [_f32p, u32, _f32p, u64, _f64p]
    lib.oracle_fir_r32_f64.restype = u64
    lib.oracle_fir_c64_f32fma.argtypes = [_f32p, u32, u32, _f32p, u64, _f32p, u64, _f32p, u32, u32]
    lib.oracle_fir_c64_f32fma.restype = u64
    lib.oracle_fir_c64_f32_omp.argtypes = [_f32p, u32, u32, _f32p, u64, _f32p, i32]
    lib.oracle_fir_c64_f32_omp.restype = u64
    lib.oracle_bpf_design_complex.argtypes = [_f32p, u32, ctypes.c_double, ctypes.c_double, u32]
    lib.oracle_bpf_design_complex.restype = ctypes.c_int
    lib.oracle_fir_c64_ctaps_f64.argtypes = [_f32p, u32, u32, _f32p, u64, _f32p, u64, _f64p]
    lib.oracle_fir_c64_ctaps_f64.restype = u64
    lib.oracle_fir_c64_nco_f64.argtypes = [_f32p, u32, u32, u32, _f32p, u64, _f32p, u64, u32, _f64p]
    lib.oracle_fir_c64_nco_f64.restype = u64
    lib.oracle_fir_c64_ctaps_f32fma.argtypes = [_f32p, u32, u32, _f32p, u64, _f32p, u64, _f32p, u32]
    lib.oracle_fir_c64_ctaps_f32fma.restype = u64
    lib.oracle_max_threads.restype = i32
    lib.oracle_err_metrics.argtypes = [_f32p, _f64p, u64, _f64p]
    del u8
    return lib


def lib(path=None):
    global _LIB
    if path is not None:
        return _bind(ctypes.CDLL(path))
    if _LIB is None:
        _LIB = _bind(ctypes.CDLL(build()))
    return _LIB


def _p32(a):
    return a.ctypes.data_as(_f32p) if a is not None else ctypes.cast(None, _f32p)


def _p64(a):
    return a.ctypes.data_as(_f64p)


def _iq(a):
    """Accept complex64 or interleaved float32; return contiguous interleaved float32."""
    a = np.asarray(a)
    if np.iscomplexobj(a):
        a = np.ascontiguousarray(a.astype(np.complex64)).view(np.float32)
    return np.ascontiguousarray(a, dtype=np.float32).reshape(-1)


WINDOWS = {"rect": 0, "hamming": 1, "hann": 2, "blackman": 3}


def bpf_design(taps, f_low=0.15, f_high=0.25, window="blackman"):
    h = np.zeros(taps, dtype=np.float32)
    ok = lib().oracle_bpf_design(_p32(h), taps, float(f_low), float(f_high), WINDOWS[window])
    if not ok:
        raise ValueError("oracle_bpf_design rejected the arguments")
    return h


def bpf_design_numpy(taps, f_low=0.15, f_high=0.25, window="blackman"):
    """Independent numpy restatement of SPEC §4 (cross-check of the C designer)."""
    n = np.arange(taps, dtype=np.float64)
    m = (taps - 1) / 2.0
    t = n - m
    a = 2.0 * np.pi * n / (taps - 1)
    w = {"rect": np.ones(taps), "hamming": 0.54 - 0.46 * np.cos(a), "hann": 0.5 - 0.5 * np.cos(a),
         "blackman": 0.42 - 0.5 * np.cos(a) + 0.08 * np.cos(2 * a)}[window]
    g = (2 * f_high * np.sinc(2 * f_high * t) - 2 * f_low * np.sinc(2 * f_low * t)) * w
    gain = np.sum(g * np.cos(2 * np.pi * 0.5 * (f_low + f_high) * t))
    return (g / gain).astype(np.float32)


def synth_iq(n, channel=0, first=0):
    x = np.empty(2 * n, dtype=np.float32)
    lib().oracle_synth_iq(_p32(x), first, n, channel)
    return x


def synth_iq_numpy(n, channel=0, first=0):
    """Independent numpy restatement of SPEC §5."""
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


def out_count(consumed, n, decim):
    return int(lib().oracle_out_count(consumed, n, decim))


def fir_f64(taps, x, decim=1, hist=None, consumed=0):
    """THE oracle (float64 accumulate). x: interleaved float32 / complex64. Returns interleaved float64."""
    taps = np.ascontiguousarray(taps, dtype=np.float32)
    x = _iq(x)
    n = x.size // 2
    hist = None if hist is None else _iq(hist)
    m = out_count(consumed, n, decim)
    y = np.empty(2 * m, dtype=np.float64)
    got = lib().oracle_fir_c64_f64(_p32(taps), taps.size, decim, _p32(hist), consumed, _p32(x), n, _p64(y))
    assert got == m
    return y


def fir_real_f64(taps, x):
    taps = np.ascontiguousarray(taps, dtype=np.float32)
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.empty(x.size, dtype=np.float64)
    lib().oracle_fir_r32_f64(_p32(taps), taps.size, _p32(x), x.size, _p64(y))
    return y


def fir_f32fma(taps, x, decim=1, hist=None, consumed=0, seg_mode=0, seg_len=0):
    """Bit-exact model of the HIP kernels' float32 fma order (SPEC §3)."""
    taps = np.ascontiguousarray(taps, dtype=np.float32)
    x = _iq(x)
    n = x.size // 2
    hist = None if hist is None else _iq(hist)
    m = out_count(consumed, n, decim)
    y = np.empty(2 * m, dtype=np.float32)
    got = lib().oracle_fir_c64_f32fma(_p32(taps), taps.size, decim, _p32(hist), consumed, _p32(x), n, _p32(y),
                                      seg_mode, seg_len)
    assert got == m
    return y


def update_history(taps_len, hist, x):
    x = _iq(x)
    hist = None if hist is None else _iq(hist)
    out = np.zeros(2 * (taps_len - 1), dtype=np.float32)
    if taps_len > 1:
        lib().oracle_update_history(_p32(out), _p32(hist), taps_len, _p32(x), x.size // 2)
    return out


def fir_f32_omp(taps, x, decim=1, threads=0, handle=None):
    """Timed CPU baseline (float32, OpenMP). Zero history, phase 0. Returns interleaved float32."""
    taps = np.ascontiguousarray(taps, dtype=np.float32)
    x = _iq(x)
    n = x.size // 2
    t = taps.size
    buf = np.zeros(2 * (n + t - 1), dtype=np.float32)
    buf[2 * (t - 1):] = x
    m = (n + decim - 1) // decim
    y = np.empty(2 * m, dtype=np.float32)
    (handle or lib()).oracle_fir_c64_f32_omp(_p32(taps), t, decim, _p32(buf), n, _p32(y), threads)
    return y


def bpf_design_complex(taps, centre=0.2, bandwidth=0.1, window="blackman"):
    """Complex taps: low-pass prototype of two-sided bandwidth `bandwidth` shifted to `centre` (interleaved re,im)."""
    g = np.zeros(2 * taps, dtype=np.float32)
    if not lib().oracle_bpf_design_complex(_p32(g), taps, float(centre), float(bandwidth), WINDOWS[window]):
        raise ValueError("oracle_bpf_design_complex rejected the arguments")
    return g


def fir_ctaps_f64(ctaps, x, decim=1, hist=None, consumed=0):
    """float64 oracle with complex taps (interleaved re,im float32)."""
    ctaps = _iq(ctaps)
    x = _iq(x)
    n, t = x.size // 2, ctaps.size // 2
    hist = None if hist is None else _iq(hist)
    m = out_count(consumed, n, decim)
    y = np.empty(2 * m, dtype=np.float64)
    assert lib().oracle_fir_c64_ctaps_f64(_p32(ctaps), t, decim, _p32(hist), consumed, _p32(x), n, _p64(y)) == m
    return y


def nco_phase_word(freq):
    """SPEC §3.2: P = round(f * 2^32) mod 2^32."""
    return int(round(float(freq) * 4294967296.0)) % (1 << 32)


def fir_nco_f64(taps, x, decim, phase_word, hist=None, consumed=0, complex_taps=False):
    """float64 oracle of NCO mix + FIR + decimation by the definition (SPEC §3.2); hist holds UNMIXED samples."""
    taps = _iq(taps) if complex_taps else np.ascontiguousarray(taps, dtype=np.float32)
    x = _iq(x)
    n, t = x.size // 2, (taps.size // 2 if complex_taps else taps.size)
    hist = None if hist is None else _iq(hist)
    m = out_count(consumed, n, decim)
    y = np.empty(2 * m, dtype=np.float64)
    assert lib().oracle_fir_c64_nco_f64(_p32(taps), t, 1 if complex_taps else 0, decim, _p32(hist), consumed, _p32(x),
                                        n, int(phase_word), _p64(y)) == m
    return y


def fir_ctaps_f32fma(ctaps, x, decim=1, hist=None, consumed=0, seg_len=32):
    ctaps = _iq(ctaps)
    x = _iq(x)
    n, t = x.size // 2, ctaps.size // 2
    hist = None if hist is None else _iq(hist)
    m = out_count(consumed, n, decim)
    y = np.empty(2 * m, dtype=np.float32)
    assert lib().oracle_fir_c64_ctaps_f32fma(_p32(ctaps), t, decim, _p32(hist), consumed, _p32(x), n, _p32(y), seg_len) == m
    return y


def max_threads():
    return int(lib().oracle_max_threads())


def err_metrics(y, ref):
    """(‖y-ŷ‖₂/‖ŷ‖₂, max|y-ŷ|/max|ŷ|) — SPEC §3."""
    y = np.ascontiguousarray(y, dtype=np.float32).reshape(-1)
    ref = np.ascontiguousarray(ref, dtype=np.float64).reshape(-1)
    assert y.size == ref.size
    out = np.zeros(2, dtype=np.float64)
    lib().oracle_err_metrics(_p32(y), _p64(ref), y.size, _p64(out))
    return float(out[0]), float(out[1])


class OracleStream:
    """Streaming wrapper with the same state as if_fir_ctx_t (history + consumed count)."""

    def __init__(self, taps, decim=1):
        self.taps = np.ascontiguousarray(taps, dtype=np.float32)
        self.decim = decim
        self.reset()

    def reset(self):
        self.hist = np.zeros(2 * (self.taps.size - 1), dtype=np.float32)
        self.consumed = 0

    def process(self, x, kind="f64", **kw):
        x = _iq(x)
        fn = fir_f64 if kind == "f64" else fir_f32fma
        y = fn(self.taps, x, self.decim, self.hist, self.consumed, **kw)
        self.hist = update_history(self.taps.size, self.hist, x)
        self.consumed += x.size // 2
        return y
