"""ctypes wrapper around oracle/libwb_detect_oracle.so — TEST INFRASTRUCTURE ONLY (only tests/ may import this).

PARITY PINNED: oracle/wb_detect_oracle.c restates the reference's WBSpectrumMonitor.detect_signals()
(/root/reference/software/opi-rf-manager/lib/wb_spectrum_monitor.js:36-180) and is checked bit-for-bit against vectors
captured from that routine (tests/golden/wb_detect_golden.json, tests/golden/make_wb_golden.js)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

# numpy mirrors of include/wb_detect.h (natural C layout: 13 doubles + 2 uint32 = 112 bytes; 16 + 112 = 128 bytes)
SIGNAL_FIELDS = ["full_start_freq", "full_end_freq", "full_center_freq", "full_bandwidth", "full_power",
                 "used_start_freq", "used_end_freq", "used_center_freq", "used_bandwidth", "used_power",
                 "symbolrate", "snr", "sbr"]
SIGNAL_DTYPE = np.dtype([(k, "<f8") for k in SIGNAL_FIELDS] + [("out_of_band", "<u4"), ("over_powered", "<u4")])
FRAME_DTYPE = np.dtype([("noise_power", "<f8"), ("beacon_valid", "<u4"), ("signal_count", "<u4"), ("beacon", SIGNAL_DTYPE)])
assert SIGNAL_DTYPE.itemsize == 112 and FRAME_DTYPE.itemsize == 128


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libwb_detect_oracle.so")
        src = os.path.join(_HERE, "wb_detect_oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "libwb_detect_oracle.so"], stdout=subprocess.DEVNULL)
        L = ctypes.CDLL(so)
        L.oracle_wb_detect.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
        L.oracle_wb_detect.restype = ctypes.c_int
        _LIB = L
    return _LIB


def detect(bins, max_signals=512):
    """One frame (uint16 array) -> (status, frame record, signals array truncated to the stored count)."""
    bins = np.ascontiguousarray(bins, dtype="<u2")
    frame = np.zeros(1, dtype=FRAME_DTYPE)
    sig = np.zeros(max_signals, dtype=SIGNAL_DTYPE)
    st = lib().oracle_wb_detect(bins.ctypes.data, bins.size, frame.ctypes.data, sig.ctypes.data, max_signals)
    return st, frame[0], sig[:min(int(frame[0]["signal_count"]), max_signals)]
