#!/usr/bin/env python3
"""ab_inproc.py <workload> [--libs a.so b.so ...] [--variants v ...] [--rounds R] [--reps K] — A/B timing in ONE process
(development tool).  Every (library build, tuning variant) pair gets its own context on the same resident input; the pairs are
timed alternately, R rounds of K back-to-back launches each after a common warm-up, so that clock state, temperature and the
box itself are common to all of them (sweeps of one variant after the other are dominated by the power controller's transient:
the first configuration of a process always looks 5-10 % slower).  Prints per pair the median / min / max of the round means and
the ratio to the first pair.  Variants: 0 = default; development variants of include/if_fir_debug.h (1000000 + bits) need
IF_FIR_DEBUG=1 (set here)."""
import argparse
import os
import statistics
import sys

os.environ.setdefault("IF_FIR_DEBUG", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import __graft_entry__ as g  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload", nargs="?", default="fir255_dec4_2p28")
    ap.add_argument("--libs", nargs="*", default=[None])
    ap.add_argument("--variants", nargs="*", type=int, default=[0])
    ap.add_argument("--rounds", type=int, default=12)
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--buffers", type=int, default=1,
                    help="rotate the launches over this many input/output buffer pairs (K > 1: consecutive launches never see the "
                         "same bytes, so nothing of a 2^26-sample input survives in the 256 MB memory-side cache from one launch "
                         "to the next -- a stream in service; timed with events around K x reps launches)")
    ap.add_argument("--i16", action="store_true")
    ap.add_argument("--nco", type=float, default=0.0)
    args = ap.parse_args()
    taps_n, decim, log2n, _ = bench.WORKLOADS[args.workload]
    n = 1 << log2n
    fir = g.load_pkg().if_fir
    taps = fir.bpf_design(taps_n)
    torch.cuda.set_device(0)
    x = torch.empty(2 * n, dtype=torch.float32, device="cuda")
    pairs = []
    for lib in args.libs:
        for v in args.variants:
            f = fir.IfFir(taps, decim, 0, dev=True, lib_path=lib)
            f.set_backend(fir.BACKEND_HIP_FFT)
            if args.nco:
                f.set_nco(args.nco)
            f.set_tuning(v)
            pairs.append((os.path.basename(lib) if lib else "default", v, f))
    f0 = pairs[0][2]
    f0.synth_device(x.data_ptr(), 0, n, 0)
    f0.synchronize()
    src = x
    if args.i16:
        src = (x * 32767.0).round().clamp(-32768, 32767).to(torch.int16)
        for _, _, f in pairs:
            f.set_input_format(fir.INPUT_I16)
    m = f0.out_count(n)
    y = torch.empty(2 * m, dtype=torch.float32, device="cuda")
    sums = []
    for name, v, f in pairs:       # one launch each: the outputs' checksum (a launch that skips blocks looks fast)
        y.zero_()
        torch.cuda.synchronize()   # (zero_ runs on torch's stream, the filter on the context's)
        f.reset()
        f.process_device(src.data_ptr(), y.data_ptr(), n)
        f.synchronize()
        sums.append(y.double().abs().sum().item())
    for _ in range(3):             # common warm-up: the power controller's transient
        for _, _, f in pairs:
            f.time_device(src.data_ptr(), y.data_ptr(), n, 0, args.reps)
    times = [[] for _ in pairs]
    if args.buffers > 1:
        assert not args.i16
        xs = [x] + [x.clone() for _ in range(args.buffers - 1)]
        ys = [y] + [torch.empty_like(y) for _ in range(args.buffers - 1)]
        stream = torch.cuda.Stream()
        for _, _, f in pairs:
            f.set_stream(stream.cuda_stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

        def timed(f):
            e0.record(stream)
            for r in range(args.reps):
                k = r % args.buffers
                f.process_device(xs[k].data_ptr(), ys[k].data_ptr(), n)
            e1.record(stream)
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / args.reps
    else:
        def timed(f):
            return f.time_device(src.data_ptr(), y.data_ptr(), n, 0, args.reps)
    for r in range(args.rounds):
        order = range(len(pairs)) if r % 2 == 0 else reversed(range(len(pairs)))
        for k in order:
            times[k].append(timed(pairs[k][2]))
    base = statistics.median(times[0])
    bytes_ = bench.algorithmic_bytes_per_sample(decim) * n * (0.6 if args.i16 and decim == 4 else 1.0)
    for (name, v, f), t, ck in zip(pairs, times, sums):
        med = statistics.median(t)
        print("%s %-22s variant %8d: median %.4f ms (min %.4f max %.4f over %d rounds of %d)  x%.4f  frac %.4f  checksum %.9e" %
              (args.workload, name, v, med, min(t), max(t), len(t), args.reps, med / base, bytes_ / (med * 1e-3) / 8e12, ck), flush=True)
    for _, _, f in pairs:
        f.close() if hasattr(f, "close") else None


if __name__ == "__main__":
    main()
