"""ctypes binding of the batched WB-transponder signal detector in libif_fir.so (include/wb_detect.h): the reference's
WBSpectrumMonitor.detect_signals() (opi-rf-manager/lib/wb_spectrum_monitor.js:36-180) for many frames per call.
No compute happens here and there is no CPU fallback."""
import ctypes

import numpy as np

from . import if_fir as _if_fir

EXPORTS = ["wb_detect_frames_device", "wb_detect_frames", "wb_detect_last_error"]

SIGNAL_FIELDS = ["full_start_freq", "full_end_freq", "full_center_freq", "full_bandwidth", "full_power",
                 "used_start_freq", "used_end_freq", "used_center_freq", "used_bandwidth", "used_power",
                 "symbolrate", "snr", "sbr"]
SIGNAL_DTYPE = np.dtype([(k, "<f8") for k in SIGNAL_FIELDS] + [("out_of_band", "<u4"), ("over_powered", "<u4")])
FRAME_DTYPE = np.dtype([("noise_power", "<f8"), ("beacon_valid", "<u4"), ("signal_count", "<u4"), ("beacon", SIGNAL_DTYPE)])


class WbDetectError(RuntimeError):
    pass


_bound = False


def lib():
    global _bound
    L = _if_fir.lib()
    if not _bound:
        vp, u32, i32 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int32
        L.wb_detect_frames_device.argtypes = [vp, u32, u32, vp, vp, u32, i32, vp]
        L.wb_detect_frames_device.restype = ctypes.c_uint8
        L.wb_detect_frames.argtypes = [vp, u32, u32, vp, vp, u32, i32]
        L.wb_detect_frames.restype = ctypes.c_uint8
        L.wb_detect_last_error.argtypes = []
        L.wb_detect_last_error.restype = ctypes.c_char_p
        _bound = True
    return L


def detect_frames(bins, max_signals=64, device=0):
    """wb_detect_frames(): bins = (frames, bins) uint16 array on the host -> (frames records, (frames, max_signals)
    signal records); entry k of a frame's signals is valid for k < min(signal_count, max_signals)."""
    bins = np.ascontiguousarray(bins, dtype="<u2")
    if bins.ndim != 2:
        raise WbDetectError("bins must be a (frames, bins) array")
    frames = np.zeros(bins.shape[0], dtype=FRAME_DTYPE)
    signals = np.zeros((bins.shape[0], max_signals), dtype=SIGNAL_DTYPE)
    if not lib().wb_detect_frames(bins.ctypes.data, bins.shape[0], bins.shape[1], frames.ctypes.data, signals.ctypes.data,
                                  int(max_signals), int(device)):
        raise WbDetectError(lib().wb_detect_last_error().decode())
    return frames, signals


def detect_frames_device(dev_bins, n_frames, n_bins, dev_frames, dev_signals, max_signals, device=0, stream=0):
    """wb_detect_frames_device(): raw device pointers (ints), asynchronous on `stream`."""
    if not lib().wb_detect_frames_device(ctypes.c_void_p(int(dev_bins)), int(n_frames), int(n_bins),
                                         ctypes.c_void_p(int(dev_frames)), ctypes.c_void_p(int(dev_signals)),
                                         int(max_signals), int(device), ctypes.c_void_p(int(stream))):
        raise WbDetectError(lib().wb_detect_last_error().decode())
