#!/usr/bin/env python3
"""bench_wb_detect.py — throughput of the batched WB signal detector (wb_detect_frames_device) with the frames resident in HBM,
and the oracle restatement on one host core beside it.  usage: python tests/bench_wb_detect.py [frames=262144] [bins=918]
(lives under tests/ because it calls the oracle as a checker and as the one-core CPU figure)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import __graft_entry__ as g  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_wb_detect import _random_frames  # noqa: E402  (the frame generator of the tests)


def main():
    n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
    n_bins = int(sys.argv[2]) if len(sys.argv) > 2 else 918
    wb = g.load_pkg().wb_detect
    from oracle import wb_oracle
    rng = np.random.default_rng(7)
    base = _random_frames(rng, 4096, n_bins)
    bins = torch.from_numpy(np.tile(base, (n_frames // 4096 + 1, 1))[:n_frames].astype(np.int16)).cuda()
    cap = 16
    frames = torch.zeros(n_frames * wb.FRAME_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    sigs = torch.zeros(n_frames * cap * wb.SIGNAL_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.Stream()
    args = (bins.data_ptr(), n_frames, n_bins, frames.data_ptr(), sigs.data_ptr(), cap, 0, stream.cuda_stream)
    torch.cuda.synchronize()
    for _ in range(3):
        wb.detect_frames_device(*args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(10):
        wb.detect_frames_device(*args)
    e1.record(stream)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    got = np.frombuffer(frames.cpu().numpy().tobytes(), dtype=wb.FRAME_DTYPE)
    t0 = time.perf_counter()
    same = True
    for k in range(2048):
        st, of, _ = wb_oracle.detect(base[k], max_signals=cap)
        same = same and got[k].tobytes() == of.tobytes()
    cpu_s = (time.perf_counter() - t0) / 2048
    print(json.dumps({"workload": "%d frames x %d bins" % (n_frames, n_bins), "kernel_ms": round(ms, 4),
                      "frames_per_s": round(n_frames / ms * 1e3), "input_gbs": round(2.0 * n_bins * n_frames / ms / 1e6, 1),
                      "hbm_frac": round(2.0 * n_bins * n_frames / ms / 1e6 / 8000.0, 4),
                      "cpu_oracle_frames_per_s_one_core": round(1.0 / cpu_s), "bit_identical_to_oracle": bool(same)}))


if __name__ == "__main__":
    main()
