#!/usr/bin/env python3
"""fbank_bench.py — uniform filter bank (if_fir_channelizer_process_device) against the same channels
computed one at a time (if_fir_set_nco contexts): time per pass over a 2^log2n-sample wideband stream, whole-output
comparison of every channel, one JSON line.
usage: python tools/fbank_bench.py [channels=8] [log2n=28] [taps=255] [decimation=4] [freq] [tuning=N] [nco=F]   (decimation 4: 4x oversampled fs/16
channels; 16: the channel rate, all 16 slots from one forward transform, round 3; "freq" (decimation 4, 8 or 16, round 4): the channels sit
at arbitrary centres on the fs/4096 grid -- if_fir_channelizer_process_device_freq -- instead of on slots)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import __graft_entry__ as g  # noqa: E402


def main():
    nch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    log2n = int(sys.argv[2]) if len(sys.argv) > 2 else 28
    taps_n = int(sys.argv[3]) if len(sys.argv) > 3 else 255
    dec = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    freq = "freq" in sys.argv[5:]
    tuning = [int(a[7:]) for a in sys.argv[5:] if a.startswith("tuning=")]   # e.g. tuning=1004096: decimation 8 without the all-slots form
    nco = ([float(a[4:]) for a in sys.argv[5:] if a.startswith("nco=")] or [0.0])[0]   # the context's NCO: a common offset of the slot grid (decimation 8 / 16)
    n = 1 << log2n
    if tuning:
        os.environ["IF_FIR_DEBUG"] = "1"
    fir = g.load_pkg().if_fir
    torch.cuda.set_device(0)
    taps = fir.bpf_design(taps_n, 0.0, 0.03 if dec == 4 else 0.02)
    slots = [(2 * c + 1) % 16 for c in range(nch)] if nch <= 8 else list(range(nch))
    # arbitrary centres: near the slots, 37 + 11 c bins of fs/4096 off them
    centres = [(((256 * s + 37 + 11 * c) + 2048) % 4096 - 2048) / 4096.0 for c, s in enumerate(slots)]

    def bank(f, ptrs):
        if freq:
            return f.channelizer_process_device_freq(centres, x.data_ptr(), ptrs, n)
        return f.channelizer_process_device(slots, x.data_ptr(), ptrs, n)
    x = torch.empty(2 * n, dtype=torch.float32, device="cuda")
    with fir.IfFir(taps, dec, 0, dev=True) as f:
        if tuning:
            f.set_tuning(tuning[0])
        if nco:
            f.set_nco(nco)
        m = f.out_count(n)
        outs = [torch.empty(2 * m, dtype=torch.float32, device="cuda") for _ in range(nch)]
        torch.cuda.synchronize()
        f.synth_device(x.data_ptr(), 0, n, 0)
        f.synchronize()
        stream = torch.cuda.Stream()
        f.set_stream(stream.cuda_stream)
        ptrs = [o.data_ptr() for o in outs]
        # ~150 ms of its own launches first: the power management settles only then (launches 5-25 after idle run 10-15 % slow;
        # round 4 until batch 10 timed exactly those: 0.906 ms for the 8 channels that take 0.81 ms settled)
        for _ in range(150):
            bank(f, ptrs)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        steps = 40
        e0.record(stream)
        for _ in range(steps):
            bank(f, ptrs)
        e1.record(stream)
        torch.cuda.synchronize()
        ms_bank = e0.elapsed_time(e1) / steps
        # the same stream position for the check: one more pass from a reset context
        f.reset()
        bank(f, ptrs)
        f.synchronize()
    # one channel at a time (what the filter bank replaces)
    ref = torch.empty(2 * m, dtype=torch.float32, device="cuda")
    worst = 0.0
    ms_single = 0.0
    for c, s in enumerate(slots):
        with fir.IfFir(taps, dec, 0, dev=True) as f1:
            fs = centres[c] if freq else (s / 16.0 if s <= 8 else s / 16.0 - 1.0) + nco   # slots above 8 are negative frequencies
            f1.set_nco(fs - 1.0 if fs > 0.5 else fs + 1.0 if fs < -0.5 else fs)
            f1.set_stream(stream.cuda_stream)
            f1.process_device(x.data_ptr(), ref.data_ptr(), n)
            f1.synchronize()
            scale = ref.abs().max().item()
            worst = max(worst, (ref - outs[c]).abs().max().item() / scale)
            for _ in range(60):
                f1.process_device(x.data_ptr(), ref.data_ptr(), n)
            e0.record(stream)
            for _ in range(20):
                f1.process_device(x.data_ptr(), ref.data_ptr(), n)
            e1.record(stream)
            torch.cuda.synchronize()
            ms_single += e0.elapsed_time(e1) / 20
    bytes_alg = (8.0 + nch * 8.0 / dec) * n   # one read of the wideband stream + every channel's output
    print(json.dumps({
        "workload": "%s: %d channels x (%d-tap prototype, decimate-by-%d) from one 2^%d-sample stream" %
                    ("filter bank, channels at arbitrary centres (fs/4096 grid)" if freq else "uniform filter bank", nch, taps_n, dec, log2n),
        "centres": [round(c, 6) for c in centres] if freq else None,
        "bytes_per_input_sample": 8.0 + nch * 8.0 / dec,
        "slots": slots, "filter_bank_ms": round(ms_bank, 4), "one_channel_at_a_time_ms": round(ms_single, 4),
        "speedup": round(ms_single / ms_bank, 2),
        "input_msamples_per_s": round(n / ms_bank / 1e3, 1), "channel_output_msamples_per_s": round(nch * m / ms_bank / 1e3, 1),
        "algorithmic_bytes": bytes_alg, "hbm_gbs": round(bytes_alg / (ms_bank * 1e-3) / 1e9, 1),
        "hbm_frac": round(bytes_alg / (ms_bank * 1e-3) / 1e9 / 8000.0, 4),
        "max_rel_diff_vs_nco_contexts": worst, "bound": 2e-6, "ok": bool(worst <= 2e-6)}))


if __name__ == "__main__":
    main()
