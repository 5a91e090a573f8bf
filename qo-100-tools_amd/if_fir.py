"""ctypes binding of libif_fir.so — one Python method per C entry point of include/if_fir.h (same names, same
argument meaning, 1/0 status turned into IfFirError).  No compute happens here.

Two libraries: libif_fir.so is the product (include/if_fir.h); libif_fir_dev.so is the same code with the development
hooks of include/if_fir_debug.h compiled in (diagnostic tuning variants, stamps, the host-only table / schedule / plan
dumps, if_fir_time_device).  `IfFir(..., dev=True)` / `IfFirMc(..., dev=True)` and the module-level debug_* helpers use
the development library; everything else uses the product."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libif_fir.so")
DEV_LIB_PATH = os.path.join(_HERE, "libif_fir_dev.so")

INPUT_F32, INPUT_I16 = 0, 1
BACKEND_AUTO, BACKEND_HIP_DIRECT, BACKEND_HIP_TAPSPLIT, BACKEND_HIP_GENERIC, BACKEND_HIP_FFT = range(5)
WINDOW_RECT, WINDOW_HAMMING, WINDOW_HANN, WINDOW_BLACKMAN = range(4)

# every symbol include/if_fir.h declares (tests check the library exports all of them)
EXPORTS = [
    "if_bpf_design", "if_bpf_design_complex", "if_fir_init", "if_fir_init_complex", "if_fir_destroy", "if_fir_reset", "if_fir_set_backend", "if_fir_get_backend",
    "if_fir_set_tuning", "if_fir_set_input_format", "if_fir_set_stream", "if_fir_synchronize", "if_fir_last_error", "if_fir_out_count",
    "if_fir_process", "if_fir_process_device", "if_fir_synth_device", "if_fir_dev_alloc",
    "if_fir_dev_free", "if_fir_dev_upload", "if_fir_dev_download", "if_fir_device_info",
    "if_fir_set_nco", "if_fir_get_nco", "if_fir_channelizer_process_device", "if_fir_channelizer_process_device_freq", "if_fir_host_alloc", "if_fir_host_free",
    "if_fir_power_device",
    "if_fir_mc_owner", "if_fir_mc_unique_id", "if_fir_mc_init", "if_fir_mc_destroy", "if_fir_mc_reset",
    "if_fir_mc_set_input_format", "if_fir_mc_process_device", "if_fir_mc_channel_ctx", "if_fir_mc_last_error",
    "if_fir_mc_set_chunk_samples", "if_fir_mc_get_chunk_samples",
]
# every symbol include/if_fir_debug.h declares: exported by libif_fir_dev.so only
DEV_EXPORTS = ["if_fir_time_device", "if_fir_debug_stamps", "if_fir_debug_fft_tables", "if_fir_debug_fft_tables_odd", "if_fir_debug_fft_tables_bank",
               "if_fir_debug_bank_plan", "if_fir_debug_bank_tail", "if_fir_debug_fft_schedule",
               "if_fir_mc_debug_plan", "if_fir_debug_queue_faults"]
MC_ID_BYTES = 128


class IfFirError(RuntimeError):
    pass


_libs = {}


def lib():
    """Load libif_fir.so, the product (raises if it has not been built: there is no fallback)."""
    return _load(LIB_PATH, False)


def dev_lib():
    """Load libif_fir_dev.so: the product's code plus the development hooks of include/if_fir_debug.h."""
    return _load(DEV_LIB_PATH, True)


def _load(path, dev):
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise IfFirError("%s is not built (%s); run `python -c 'import __graft_entry__ as g; g.build()'`"
                         % (os.path.basename(path), path))
    L = ctypes.CDLL(path)
    u8, u32, u64, i32, vp = ctypes.c_uint8, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int32, ctypes.c_void_p
    f32p = ctypes.POINTER(ctypes.c_float)
    L.if_bpf_design.argtypes = [f32p, u32, ctypes.c_double, ctypes.c_double, u32]
    L.if_bpf_design.restype = u8
    L.if_fir_init.argtypes = [ctypes.POINTER(vp), f32p, u32, u32, u64, i32]
    L.if_fir_init.restype = u8
    L.if_fir_init_complex.argtypes = [ctypes.POINTER(vp), f32p, u32, u32, u64, i32]
    L.if_fir_init_complex.restype = u8
    L.if_bpf_design_complex.argtypes = [f32p, u32, ctypes.c_double, ctypes.c_double, u32]
    L.if_bpf_design_complex.restype = u8
    L.if_fir_destroy.argtypes = [vp]
    L.if_fir_destroy.restype = None
    L.if_fir_reset.argtypes = [vp]
    L.if_fir_reset.restype = u8
    L.if_fir_set_backend.argtypes = [vp, u32]
    L.if_fir_set_backend.restype = u8
    L.if_fir_get_backend.argtypes = [vp]
    L.if_fir_get_backend.restype = u32
    L.if_fir_set_input_format.argtypes = [vp, u32]
    L.if_fir_set_input_format.restype = u8
    L.if_fir_set_tuning.argtypes = [vp, u32]
    L.if_fir_set_tuning.restype = u8
    L.if_fir_set_stream.argtypes = [vp, vp]
    L.if_fir_set_stream.restype = u8
    L.if_fir_synchronize.argtypes = [vp]
    L.if_fir_synchronize.restype = u8
    L.if_fir_last_error.argtypes = [vp]
    L.if_fir_last_error.restype = ctypes.c_char_p
    L.if_fir_out_count.argtypes = [vp, u64]
    L.if_fir_out_count.restype = u64
    L.if_fir_process.argtypes = [vp, f32p, f32p, u64, ctypes.POINTER(u64)]
    L.if_fir_process.restype = u8
    L.if_fir_process_device.argtypes = [vp, vp, vp, u64, ctypes.POINTER(u64)]
    L.if_fir_process_device.restype = u8
    L.if_fir_synth_device.argtypes = [vp, vp, u64, u64, u32]
    L.if_fir_synth_device.restype = u8
    L.if_fir_dev_alloc.argtypes = [vp, ctypes.POINTER(vp), u64]
    L.if_fir_dev_alloc.restype = u8
    L.if_fir_dev_free.argtypes = [vp, vp]
    L.if_fir_dev_free.restype = u8
    L.if_fir_dev_upload.argtypes = [vp, vp, vp, u64]
    L.if_fir_dev_upload.restype = u8
    L.if_fir_dev_download.argtypes = [vp, vp, vp, u64]
    L.if_fir_dev_download.restype = u8
    L.if_fir_device_info.argtypes = [vp, ctypes.c_char_p, u32]
    L.if_fir_device_info.restype = u8
    L.if_fir_set_nco.argtypes = [vp, ctypes.c_double]
    L.if_fir_set_nco.restype = u8
    L.if_fir_get_nco.argtypes = [vp, ctypes.POINTER(ctypes.c_double)]
    L.if_fir_get_nco.restype = u8
    L.if_fir_channelizer_process_device.argtypes = [vp, u32, ctypes.POINTER(u32), vp, ctypes.POINTER(vp), u64,
                                                    ctypes.POINTER(u64)]
    L.if_fir_channelizer_process_device.restype = u8
    L.if_fir_channelizer_process_device_freq.argtypes = [vp, u32, ctypes.POINTER(ctypes.c_double), vp, ctypes.POINTER(vp), u64,
                                                         ctypes.POINTER(u64)]
    L.if_fir_channelizer_process_device_freq.restype = u8
    L.if_fir_host_alloc.argtypes = [vp, ctypes.POINTER(vp), u64]
    L.if_fir_host_alloc.restype = u8
    L.if_fir_host_free.argtypes = [vp, vp]
    L.if_fir_host_free.restype = u8
    L.if_fir_power_device.argtypes = [vp, vp, u64, ctypes.POINTER(ctypes.c_double)]
    L.if_fir_power_device.restype = u8
    u8p = ctypes.POINTER(ctypes.c_uint8)
    L.if_fir_mc_owner.argtypes = [u32, u32]
    L.if_fir_mc_owner.restype = u32
    L.if_fir_mc_unique_id.argtypes = [u8p]
    L.if_fir_mc_unique_id.restype = u8
    L.if_fir_mc_init.argtypes = [ctypes.POINTER(vp), u32, f32p, u32, u32, u64, i32, u32, u32, u8p]
    L.if_fir_mc_init.restype = u8
    L.if_fir_mc_destroy.argtypes = [vp]
    L.if_fir_mc_destroy.restype = None
    L.if_fir_mc_reset.argtypes = [vp]
    L.if_fir_mc_reset.restype = u8
    L.if_fir_mc_set_input_format.argtypes = [vp, u32]
    L.if_fir_mc_set_input_format.restype = u8
    L.if_fir_mc_process_device.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(vp), u64, ctypes.POINTER(u64)]
    L.if_fir_mc_process_device.restype = u8
    L.if_fir_mc_channel_ctx.argtypes = [vp, u32]
    L.if_fir_mc_channel_ctx.restype = vp
    L.if_fir_mc_last_error.argtypes = [vp]
    L.if_fir_mc_last_error.restype = ctypes.c_char_p
    L.if_fir_mc_set_chunk_samples.argtypes = [vp, u64]
    L.if_fir_mc_set_chunk_samples.restype = u8
    L.if_fir_mc_get_chunk_samples.argtypes = [vp, ctypes.POINTER(u64), ctypes.POINTER(u64)]
    L.if_fir_mc_get_chunk_samples.restype = u8
    if dev:
        L.if_fir_time_device.argtypes = [vp, vp, vp, u64, u32, u32, f32p]
        L.if_fir_time_device.restype = u8
        L.if_fir_debug_stamps.argtypes = [vp, ctypes.POINTER(u64), u32]
        L.if_fir_debug_stamps.restype = u32
        L.if_fir_debug_fft_tables.argtypes = [f32p, u32, u32, u32, u32, f32p, u32]
        L.if_fir_debug_fft_tables.restype = u32
        L.if_fir_debug_fft_schedule.argtypes = [u64, u32, ctypes.POINTER(ctypes.c_int64)]
        L.if_fir_debug_fft_schedule.restype = u8
        L.if_fir_mc_debug_plan.argtypes = [u32, u32, u32, u64, u32, u32, u32, u64, u64, ctypes.POINTER(u64), u32]
        L.if_fir_mc_debug_plan.restype = u32
        L.if_fir_debug_queue_faults.argtypes = [vp, ctypes.POINTER(u32)]
        L.if_fir_debug_queue_faults.restype = u8
    _libs[path] = L
    return L


def _f32p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def bpf_design(taps, f_low=0.15, f_high=0.25, window=WINDOW_BLACKMAN):
    """if_bpf_design(): windowed-sinc band-pass taps (float32)."""
    h = np.zeros(max(int(taps), 0), dtype=np.float32)
    if not lib().if_bpf_design(_f32p(h), int(taps), float(f_low), float(f_high), int(window)):
        raise IfFirError("if_bpf_design rejected taps=%r band=[%r,%r] window=%r" % (taps, f_low, f_high, window))
    return h


def bpf_design_complex(taps, centre=0.2, bandwidth=0.1, window=WINDOW_BLACKMAN):
    """if_bpf_design_complex(): channel-selection taps, interleaved (re, im) float32."""
    g = np.zeros(2 * max(int(taps), 0), dtype=np.float32)
    if not lib().if_bpf_design_complex(_f32p(g), int(taps), float(centre), float(bandwidth), int(window)):
        raise IfFirError("if_bpf_design_complex rejected taps=%r centre=%r bandwidth=%r" % (taps, centre, bandwidth))
    return g


class IfFir:
    """One if_fir_ctx_t.  Methods mirror the C entry points."""

    def __init__(self, taps, decimation=1, max_samples=1 << 20, device=0, backend=None, complex_taps=False, dev=False,
                 lib_path=None):
        # lib_path (development tools only, tools/ab_inproc.py): another build of the development library, loaded beside the
        # default one so that two builds can be timed alternately in one process
        self._L = _load(os.path.abspath(lib_path), True) if lib_path else dev_lib() if dev else lib()
        taps = np.asarray(taps)
        if np.iscomplexobj(taps):
            taps = np.ascontiguousarray(taps.astype(np.complex64)).view(np.float32)
            complex_taps = True
        taps = np.ascontiguousarray(taps, dtype=np.float32)
        self._ctx = ctypes.c_void_p()
        self.taps = taps
        self.decimation = int(decimation)
        if complex_taps:
            ok = self._L.if_fir_init_complex(ctypes.byref(self._ctx), _f32p(taps), taps.size // 2, self.decimation,
                                           int(max_samples), int(device))
        else:
            ok = self._L.if_fir_init(ctypes.byref(self._ctx), _f32p(taps), taps.size, self.decimation,
                                   int(max_samples), int(device))
        if not ok:
            self._ctx = ctypes.c_void_p()
            raise IfFirError(self._L.if_fir_last_error(None).decode())
        if backend is not None:
            self.set_backend(backend)

    def _check(self, ok):
        if not ok:
            raise IfFirError(self._L.if_fir_last_error(self._ctx).decode())

    def close(self):
        if self._ctx:
            self._L.if_fir_destroy(self._ctx)
            self._ctx = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def reset(self):
        self._check(self._L.if_fir_reset(self._ctx))

    def set_backend(self, backend):
        self._check(self._L.if_fir_set_backend(self._ctx, int(backend)))

    def get_backend(self):
        return int(self._L.if_fir_get_backend(self._ctx))

    def set_input_format(self, fmt):
        self._check(self._L.if_fir_set_input_format(self._ctx, int(fmt)))
        self._i16 = (int(fmt) == INPUT_I16)

    def set_nco(self, freq):
        """if_fir_set_nco(): mix the input with exp(-j 2 pi f a) ahead of the filter (SPEC §3.2); 0 = off."""
        self._check(self._L.if_fir_set_nco(self._ctx, float(freq)))

    def get_nco(self):
        f = ctypes.c_double(0.0)
        self._check(self._L.if_fir_get_nco(self._ctx, ctypes.byref(f)))
        return float(f.value)

    def power_device(self, dev_iq, samples):
        """if_fir_power_device(): mean(|y|^2) of a device IQ buffer."""
        p = ctypes.c_double(0.0)
        self._check(self._L.if_fir_power_device(self._ctx, ctypes.c_void_p(int(dev_iq)), int(samples), ctypes.byref(p)))
        return float(p.value)

    def host_alloc(self, count, dtype=np.float32):
        """if_fir_host_alloc(): a page-locked numpy array of `count` elements (free it with host_free(array))."""
        dtype = np.dtype(dtype)
        p = ctypes.c_void_p(None)
        self._check(self._L.if_fir_host_alloc(self._ctx, ctypes.byref(p), int(count) * dtype.itemsize))
        buf = (ctypes.c_char * (int(count) * dtype.itemsize)).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dtype, count=int(count))
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[arr.ctypes.data] = p.value
        return arr

    def host_free(self, arr):
        p = self._pinned.pop(arr.ctypes.data)
        self._check(self._L.if_fir_host_free(self._ctx, ctypes.c_void_p(p)))

    def process_into(self, iq, out):
        """if_fir_process() with caller-provided host arrays (e.g. from host_alloc); returns the output sample count."""
        n = iq.size // 2
        m = ctypes.c_uint64(0)
        self._check(self._L.if_fir_process(self._ctx, ctypes.cast(iq.ctypes.data, ctypes.POINTER(ctypes.c_float)),
                                         _f32p(out), n, ctypes.byref(m)))
        return int(m.value)

    def channelizer_process_device_freq(self, centres, dev_in, dev_outs, samples):
        """if_fir_channelizer_process_device_freq(): channel c centred at centres[c] cycles/sample (decimation 4, 8, 12, ..., 64)."""
        k = len(centres)
        fc = (ctypes.c_double * k)(*[float(v) for v in centres])
        po = (ctypes.c_void_p * k)(*[ctypes.c_void_p(int(p)) for p in dev_outs])
        m = ctypes.c_uint64(0)
        self._check(self._L.if_fir_channelizer_process_device_freq(self._ctx, k, fc, ctypes.c_void_p(int(dev_in)), po,
                                                                   int(samples), ctypes.byref(m)))
        return int(m.value)

    def channelizer_process_device(self, slots, dev_in, dev_outs, samples):
        """if_fir_channelizer_process_device(): uniform filter bank, channel c mixed down by slots[c]/16."""
        k = len(slots)
        sl = (ctypes.c_uint32 * k)(*[int(v) for v in slots])
        po = (ctypes.c_void_p * k)(*[ctypes.c_void_p(int(p)) for p in dev_outs])
        m = ctypes.c_uint64(0)
        self._check(self._L.if_fir_channelizer_process_device(self._ctx, k, sl, ctypes.c_void_p(int(dev_in)), po,
                                                            int(samples), ctypes.byref(m)))
        return int(m.value)

    def set_tuning(self, variant):
        self._check(self._L.if_fir_set_tuning(self._ctx, int(variant)))

    def set_stream(self, stream_handle):
        self._check(self._L.if_fir_set_stream(self._ctx, ctypes.c_void_p(stream_handle or None)))

    def synchronize(self):
        self._check(self._L.if_fir_synchronize(self._ctx))

    def out_count(self, samples):
        return int(self._L.if_fir_out_count(self._ctx, int(samples)))

    def device_info(self):
        buf = ctypes.create_string_buffer(256)
        self._check(self._L.if_fir_device_info(self._ctx, buf, 256))
        return buf.value.decode()

    def process(self, iq):
        """if_fir_process(): host interleaved float32 (or complex64) in, interleaved float32 out."""
        iq = np.asarray(iq)
        if getattr(self, "_i16", False):
            iq = np.ascontiguousarray(iq, dtype=np.int16).reshape(-1)
        else:
            if np.iscomplexobj(iq):
                iq = np.ascontiguousarray(iq.astype(np.complex64)).view(np.float32)
            iq = np.ascontiguousarray(iq, dtype=np.float32).reshape(-1)
        n = iq.size // 2
        out = np.empty(2 * self.out_count(n), dtype=np.float32)
        m = ctypes.c_uint64(0)
        dummy = np.zeros(2, dtype=np.float32)
        src = iq if n else dummy
        self._check(self._L.if_fir_process(self._ctx, ctypes.cast(src.ctypes.data, ctypes.POINTER(ctypes.c_float)),
                                         _f32p(out if out.size else dummy), n, ctypes.byref(m)))
        assert m.value * 2 == out.size
        return out

    def process_device(self, dev_in, dev_out, samples):
        """if_fir_process_device(): raw device pointers (ints), asynchronous. Returns the output sample count."""
        m = ctypes.c_uint64(0)
        self._check(self._L.if_fir_process_device(self._ctx, ctypes.c_void_p(dev_in), ctypes.c_void_p(dev_out),
                                                int(samples), ctypes.byref(m)))
        return int(m.value)

    def synth_device(self, dev_iq, first, samples, channel=0):
        self._check(self._L.if_fir_synth_device(self._ctx, ctypes.c_void_p(dev_iq), int(first), int(samples),
                                              int(channel)))

    def time_device(self, dev_in, dev_out, samples, warmup=3, reps=10):
        """if_fir_time_device() (development library: construct with dev=True)."""
        ms = ctypes.c_float(0)
        self._check(self._L.if_fir_time_device(self._ctx, ctypes.c_void_p(dev_in), ctypes.c_void_p(dev_out),
                                             int(samples), int(warmup), int(reps), ctypes.byref(ms)))
        return float(ms.value)

    def debug_queue_faults(self):
        """if_fir_debug_queue_faults() (development library): expired bounded waits of the block queue, 0 when healthy."""
        n = ctypes.c_uint32(0)
        self._check(self._L.if_fir_debug_queue_faults(self._ctx, ctypes.byref(n)))
        return int(n.value)

    def debug_stamps(self, waves=None):
        """Arm (waves=None) or fetch the per-wave diagnostic stamps of the last persistent-kernel launch."""
        if waves is None:
            self._L.if_fir_debug_stamps(self._ctx, None, 0)
            return None
        buf = np.zeros(4 * waves, dtype=np.uint64)
        n = self._L.if_fir_debug_stamps(self._ctx, buf.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), buf.size)
        return buf[:n].reshape(-1, 4)

    # device memory helpers (pure C hosts use these instead of HIP headers)
    def dev_alloc(self, nbytes):
        p = ctypes.c_void_p()
        self._check(self._L.if_fir_dev_alloc(self._ctx, ctypes.byref(p), int(nbytes)))
        return p.value

    def dev_free(self, ptr):
        self._check(self._L.if_fir_dev_free(self._ctx, ctypes.c_void_p(ptr)))

    def dev_upload(self, ptr, host):
        host = np.ascontiguousarray(host)
        self._check(self._L.if_fir_dev_upload(self._ctx, ctypes.c_void_p(ptr), host.ctypes.data_as(ctypes.c_void_p),
                                            host.nbytes))

    def dev_download(self, ptr, nbytes, dtype=np.float32):
        host = np.empty(nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        self._check(self._L.if_fir_dev_download(self._ctx, host.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(ptr),
                                              host.nbytes))
        return host


FFT_TABLE_FLOATS = 2 * (4096 + 4096 + 256 + 1024 + 1024 + 64 + 256)


def debug_fft_tables(taps, decimation, complex_taps=False, nco_delta=0):
    """if_fir_debug_fft_tables(): the overlap-save kernel's table image as complex64 sections (host-only, no GPU)."""
    taps = np.ascontiguousarray(taps, dtype=np.float32)
    t = taps.size // 2 if complex_taps else taps.size
    out = np.zeros(FFT_TABLE_FLOATS, dtype=np.float32)
    n = dev_lib().if_fir_debug_fft_tables(_f32p(taps), t, 1 if complex_taps else 0, int(decimation), int(nco_delta) & 0xFFFFFFFF,
                                      _f32p(out), out.size)
    if n != FFT_TABLE_FLOATS:
        raise IfFirError("if_fir_debug_fft_tables: (taps=%d, decimation=%d) is not served by the overlap-save kernel" % (t, decimation))
    c = out.view(np.complex64)
    return {"tw1": c[0:4096], "hp": c[4096:8192], "tw2": c[8192:8448], "twd": c[8448:9472], "twe": c[9472:10496],
            "ncob": c[10496:10560], "twf": c[10560:10816]}


def debug_fft_tables_bank(taps, bank, parity=0, complex_taps=False):
    """if_fir_debug_fft_tables_bank(): a filter bank's table image (bank 8 or 16) as complex64 sections (host-only)."""
    taps = np.ascontiguousarray(taps, dtype=np.float32)
    t = taps.size // 2 if complex_taps else taps.size
    out = np.zeros(FFT_TABLE_FLOATS, dtype=np.float32)
    L = dev_lib()
    L.if_fir_debug_fft_tables_bank.restype = ctypes.c_uint32
    n = L.if_fir_debug_fft_tables_bank(_f32p(taps), ctypes.c_uint32(t), ctypes.c_uint32(1 if complex_taps else 0), ctypes.c_uint32(int(bank)),
                                       ctypes.c_uint32(int(parity)), _f32p(out), ctypes.c_uint32(out.size))
    if n != FFT_TABLE_FLOATS:
        raise IfFirError("if_fir_debug_fft_tables_bank: (taps=%d, bank=%d) is not served" % (t, bank))
    c = out.view(np.complex64)
    return {"tw1": c[0:4096], "hp": c[4096:8192], "tw2": c[8192:8448], "twd": c[8448:9472], "twe": c[9472:10496], "ncob": c[10496:10560]}


def debug_bank_tail(decimation, own_centres=False):
    """if_fir_debug_bank_tail(): the bank's tail (4, 8, 16) for a decimation, 0 if the bank does not serve it."""
    L = dev_lib()
    L.if_fir_debug_bank_tail.restype = ctypes.c_uint32
    return int(L.if_fir_debug_bank_tail(ctypes.c_uint32(int(decimation)), ctypes.c_uint32(1 if own_centres else 0)))


def debug_bank8_plan(slots):
    """if_fir_debug_bank_plan(): (mask of the even slots' all-slots launch, mask of the odd slots', channels left per channel)."""
    k = len(slots)
    arr = (ctypes.c_uint32 * k)(*[int(v) for v in slots])
    out = (ctypes.c_uint32 * 3)()
    L = dev_lib()
    L.if_fir_debug_bank_plan.restype = ctypes.c_uint8
    if not L.if_fir_debug_bank_plan(arr, ctypes.c_uint32(k), out):
        raise IfFirError("if_fir_debug_bank_plan: 1..16 channels")
    return int(out[0]), int(out[1]), [c for c in range(k) if (out[2] >> c) & 1]


def debug_fft_tables_odd(taps, decimation, complex_taps=False, nco_delta=0):
    """if_fir_debug_fft_tables_odd(): the odd-decimation kernel's table image (decimation 3, 9, 15, ...) as complex64 sections."""
    taps = np.ascontiguousarray(taps, dtype=np.float32)
    t = taps.size // 2 if complex_taps else taps.size
    nfl = 10496 + 512
    out = np.zeros(nfl, dtype=np.float32)
    L = dev_lib()
    L.if_fir_debug_fft_tables_odd.restype = ctypes.c_uint32
    n = L.if_fir_debug_fft_tables_odd(_f32p(taps), ctypes.c_uint32(t), ctypes.c_uint32(1 if complex_taps else 0),
                                      ctypes.c_uint32(int(decimation)), ctypes.c_uint32(int(nco_delta) & 0xFFFFFFFF), _f32p(out),
                                      ctypes.c_uint32(out.size))
    if n != nfl:
        raise IfFirError("if_fir_debug_fft_tables_odd: (taps=%d, decimation=%d) is not served by the odd-decimation kernel" % (t, decimation))
    c = out.view(np.complex64)
    return {"g": c[0:3072], "tb": c[3072:3328], "tc": c[3328:4096], "twd": c[4096:5120], "twe": c[5120:5184], "ncob": c[5184:5248], "pht": c[5248:5504]}


def debug_fft_schedule(nblocks, workgroups=256):
    """if_fir_debug_fft_schedule(): block-queue layout of an overlap-save launch (host-only): RA = blocks per group,
    nA = groups, RB = static groups per workgroup, nB = 0, tickets = bound of the global counter, wgs = workgroups."""
    out = (ctypes.c_int64 * 6)()
    if not dev_lib().if_fir_debug_fft_schedule(int(nblocks), int(workgroups), out):
        raise IfFirError("if_fir_debug_fft_schedule: bad arguments")
    return dict(zip(("RA", "nA", "RB", "nB", "tickets", "wgs"), [int(v) for v in out]))


def mc_owner(channel, world):
    """if_fir_mc_owner(): the rank that filters a channel (channel mod world)."""
    return int(lib().if_fir_mc_owner(int(channel), int(world)))


# a chunk request the tests use: 215 040 samples (round 3's global unit); the library rounds any request to the context's own
# unit = lcm(block advance of its filter, 2 D) (IfFirMc.get_chunk_samples)
MC_CHUNK_UNIT = 215040
MC_NEVER_SPLIT = (1 << 64) - 1


def mc_debug_plan(world, channels, rank, samples, in_bytes=8, decimation=1, consumed=0, chunk=0, taps=255):
    """if_fir_mc_debug_plan(): list of dict(kind, phase, group, peer, channel, chunk, offset, bytes) in posting order
    (host-only; chunk=0 means one piece here)."""
    n = dev_lib().if_fir_mc_debug_plan(int(world), int(channels), int(rank), int(samples), int(in_bytes), int(taps),
                                       int(decimation), int(consumed), int(chunk), None, 0)
    buf = np.zeros(8 * max(n, 1), dtype=np.uint64)
    dev_lib().if_fir_mc_debug_plan(int(world), int(channels), int(rank), int(samples), int(in_bytes), int(taps), int(decimation),
                                   int(consumed), int(chunk), buf.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), n)
    keys = ("kind", "phase", "group", "peer", "channel", "chunk", "offset", "bytes")
    return [dict(zip(keys, (int(v) for v in buf[8 * i:8 * i + 8]))) for i in range(n)]


def mc_unique_id():
    """if_fir_mc_unique_id(): rank 0's RCCL bootstrap id (128 bytes) to hand to the other ranks."""
    buf = (ctypes.c_uint8 * MC_ID_BYTES)()
    if not lib().if_fir_mc_unique_id(buf):
        raise IfFirError(lib().if_fir_mc_last_error(None).decode())
    return bytes(buf)


class IfFirMc:
    """One if_fir_mc_ctx_t: channel c -> rank c mod world, inputs/outputs on rank 0's GPU (see include/if_fir.h)."""

    def __init__(self, taps, decimation, max_samples, device=0, rank=0, world=1, unique_id=None, dev=False):
        self._L = dev_lib() if dev else lib()
        taps = np.ascontiguousarray(np.asarray(taps, dtype=np.float32))
        if taps.ndim != 2:
            raise IfFirError("taps must be a (channels, taps) array")
        self.channels, self.rank, self.world = int(taps.shape[0]), int(rank), int(world)
        self._ctx = ctypes.c_void_p(None)
        idbuf = None
        if unique_id is not None:
            idbuf = (ctypes.c_uint8 * MC_ID_BYTES).from_buffer_copy(bytes(unique_id))
        ok = self._L.if_fir_mc_init(ctypes.byref(self._ctx), self.channels, _f32p(taps), int(taps.shape[1]),
                                  int(decimation), int(max_samples), int(device), self.rank, self.world, idbuf)
        if not ok:
            self._ctx = ctypes.c_void_p(None)
            raise IfFirError(self._L.if_fir_mc_last_error(None).decode())

    def _check(self, ok):
        if not ok:
            raise IfFirError(self._L.if_fir_mc_last_error(self._ctx).decode())

    def reset(self):
        self._check(self._L.if_fir_mc_reset(self._ctx))

    def set_input_format(self, fmt):
        self._check(self._L.if_fir_mc_set_input_format(self._ctx, int(fmt)))

    def set_chunk_samples(self, chunk):
        """0 = default chunk, MC_NEVER_SPLIT = whole calls, else a request in samples (rounded to the context's unit)."""
        self._check(self._L.if_fir_mc_set_chunk_samples(self._ctx, int(chunk)))

    def get_chunk_samples(self):
        """(chunk in effect, 0 = calls are not split; the unit it is a multiple of)"""
        c, u = ctypes.c_uint64(0), ctypes.c_uint64(0)
        self._check(self._L.if_fir_mc_get_chunk_samples(self._ctx, ctypes.byref(c), ctypes.byref(u)))
        return int(c.value), int(u.value)

    def channel_ctx(self, channel):
        """Raw if_fir_ctx_t* (int) of a channel this rank owns, else None."""
        return self._L.if_fir_mc_channel_ctx(self._ctx, int(channel))

    def set_backend(self, backend):
        """if_fir_set_backend() on every channel this rank owns."""
        for c in range(self.channels):
            h = self.channel_ctx(c)
            if h and not self._L.if_fir_set_backend(h, int(backend)):
                raise IfFirError(self._L.if_fir_last_error(h).decode())

    def process_device(self, dev_in, dev_out, samples):
        """if_fir_mc_process_device(): lists of device pointers (ints) on rank 0, None elsewhere.  Returns the
        per-channel output sample count."""
        m = ctypes.c_uint64(0)
        pin = pout = None
        if dev_in is not None:
            pin = (ctypes.c_void_p * self.channels)(*[ctypes.c_void_p(int(p)) for p in dev_in])
            pout = (ctypes.c_void_p * self.channels)(*[ctypes.c_void_p(int(p)) for p in dev_out])
        self._check(self._L.if_fir_mc_process_device(self._ctx, pin, pout, int(samples), ctypes.byref(m)))
        return int(m.value)

    def close(self):
        if self._ctx:
            self._L.if_fir_mc_destroy(self._ctx)
            self._ctx = ctypes.c_void_p(None)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
