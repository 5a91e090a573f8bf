#!/usr/bin/env python3
"""fbank_ab.py [--libs a.so b.so ...] [--cases dec:channels:form ...] [--rounds R] [--reps K] [--log2n N] [--taps T] — filter-bank
timings of several library builds in ONE process (development tool; the filter bank's ab_inproc.py).  form = slots | freq
(channels at arbitrary centres on the fs/4096 grid, if_fir_channelizer_process_device_freq).  Every (library, case) pair gets its
own context on the same resident wideband stream; the libraries of a case are timed alternately, R rounds of K launches after a
common settling phase, and the outputs of every library are compared with the first one's (max |difference| relative to the
first's largest sample).  One line per (case, library): median / min / max of the round means, ratio to the first library,
fraction of the 8 TB/s roofline at 8 + 8 C / D bytes per input sample."""
import argparse
import os
import statistics
import sys

os.environ.setdefault("IF_FIR_DEBUG", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import __graft_entry__ as g  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", nargs="*", default=[None])
    ap.add_argument("--cases", nargs="*", default=["8:8:freq", "16:8:freq", "64:8:freq", "4:8:freq", "16:16:freq", "8:16:freq"])
    ap.add_argument("--rounds", type=int, default=8)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--log2n", type=int, default=28)
    ap.add_argument("--taps", type=int, default=255)
    ap.add_argument("--tuning", type=int, default=0)
    args = ap.parse_args()
    fir = g.load_pkg().if_fir
    torch.cuda.set_device(0)
    n = 1 << args.log2n
    x = torch.empty(2 * n, dtype=torch.float32, device="cuda")
    stream = torch.cuda.Stream()
    first = True
    for case in args.cases:
        dec, nch, form = case.split(":")
        dec, nch = int(dec), int(nch)
        freq = form == "freq"
        taps = fir.bpf_design(args.taps, 0.0, 0.03 if dec == 4 else 0.02)
        slots = [(2 * c + 1) % 16 for c in range(nch)] if nch <= 8 else list(range(nch))
        centres = [(((256 * s + 37 + 11 * c) + 2048) % 4096 - 2048) / 4096.0 for c, s in enumerate(slots)]
        ctxs = []
        for lib in args.libs:
            f = fir.IfFir(taps, dec, 0, dev=True, lib_path=lib)
            if args.tuning:
                f.set_tuning(args.tuning)
            f.set_stream(stream.cuda_stream)
            ctxs.append((os.path.basename(lib) if lib else "default", f))
        if first:
            ctxs[0][1].synth_device(x.data_ptr(), 0, n, 0)
            ctxs[0][1].synchronize()
            first = False
        m = ctxs[0][1].out_count(n)
        outs = [torch.empty(2 * m, dtype=torch.float32, device="cuda") for _ in range(nch)]
        ptrs = [o.data_ptr() for o in outs]

        def bank(f):
            if freq:
                return f.channelizer_process_device_freq(centres, x.data_ptr(), ptrs, n)
            return f.channelizer_process_device(slots, x.data_ptr(), ptrs, n)
        # outputs of every library against the first one's
        ref, diffs = None, []
        for name, f in ctxs:
            for o in outs:
                o.zero_()
            torch.cuda.synchronize()   # (zero_ runs on torch's stream, the bank on its own non-blocking one)
            f.reset()
            bank(f)
            f.synchronize()
            torch.cuda.synchronize()
            if os.environ.get("FBANK_AB_CHECKSUMS"):
                for cc in (0, 3, 7):
                    if cc < nch:
                        print("  samples ch%d %s: %s" % (cc, name, outs[cc][2000:2008].tolist()), flush=True)
                print("  checksums %s %s: %s" % (case, name, ["%.6e" % o.double().abs().sum().item() for o in outs]), flush=True)
            if ref is None:
                ref = [o.clone() for o in outs]
                diffs.append(0.0)
            else:
                per = [((o - q).abs().max() / q.abs().max()).item() for o, q in zip(outs, ref)]
                diffs.append(max(per))
                if max(per) > 1e-5:
                    print("  per-channel diff of %s: %s; zero outputs: %s / %s" % (name, ["%.2e" % v for v in per],
                          [int((o == 0).all().item()) for o in outs], [int((q == 0).all().item()) for q in ref]), flush=True)
        del ref
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

        def timed(f):
            e0.record(stream)
            for _ in range(args.reps):
                bank(f)
            e1.record(stream)
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / args.reps
        for _ in range(3):          # settle: the power controller's transient (about 150 ms of the bank's own launches)
            for _, f in ctxs:
                timed(f)
        times = [[] for _ in ctxs]
        for r in range(args.rounds):
            order = range(len(ctxs)) if r % 2 == 0 else reversed(range(len(ctxs)))
            for k in order:
                times[k].append(timed(ctxs[k][1]))
        base = statistics.median(times[0])
        bytes_alg = (8.0 + nch * 8.0 / dec) * n
        for (name, f), t, d in zip(ctxs, times, diffs):
            med = statistics.median(t)
            print("bank %-10s %-24s median %.4f ms (min %.4f max %.4f over %d rounds of %d)  x%.4f  frac %.4f  max diff vs first %.2e" %
                  (case, name, med, min(t), max(t), len(t), args.reps, med / base, bytes_alg / (med * 1e-3) / 8e12, d), flush=True)
            f.close() if hasattr(f, "close") else None


if __name__ == "__main__":
    main()
