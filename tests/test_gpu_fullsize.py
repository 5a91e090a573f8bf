"""Full-size (BASELINE.json configs[1], configs[2]) property tests on the GPU: the oracle cannot run 2^28 samples in
seconds, so parity is checked (a) exactly on windows of the stream against the oracle, (b) through size-independent
properties: one call ≡ two calls at an odd split (tile placement independence, history/phase carry), and a checksum
of the whole output that must not depend on the split."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SEG = dict(seg_mode=1, seg_len=32)


@pytest.mark.parametrize("t,d,log2n", [(127, 1, 26), (255, 4, 28), (255, 1, 26)])
def test_full_size_windows_and_split_invariance(fir, oracle, gpu_ok, t, d, log2n):
    import torch
    torch.cuda.set_device(0)
    n = 1 << log2n
    taps = fir.bpf_design(t)
    with fir.IfFir(taps, d, 0, backend=fir.BACKEND_HIP_DIRECT) as f:
        x = torch.empty(2 * n, dtype=torch.float32, device="cuda")
        m = f.out_count(n)
        y = torch.empty(2 * m, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        f.synth_device(x.data_ptr(), 0, n, 0)
        assert f.process_device(x.data_ptr(), y.data_ptr(), n) == m
        f.synchronize()
        # (a) windows: start, tile seams, an arbitrary interior place, the very end
        w = 4096
        for start in [0, 8192 - 100, (n // 3) & ~3, n - w]:
            lo = max(0, start - (t - 1))
            xs = x[2 * lo:2 * (start + w)].cpu().numpy()
            assert np.array_equal(xs, oracle.synth_iq(start + w - lo, 0, lo))
            hist = np.zeros(2 * (t - 1), dtype=np.float32)
            hist[2 * (t - 1 - (start - lo)):] = xs[:2 * (start - lo)]
            model = oracle.fir_f32fma(taps, xs[2 * (start - lo):], d, hist, start, **SEG)
            first_out = (start + d - 1) // d
            got = y[2 * first_out:2 * first_out + model.size].cpu().numpy()
            assert np.array_equal(got, model), (start, np.max(np.abs(got - model)))
            ref = oracle.fir_f64(taps, xs[2 * (start - lo):], d, hist, start)
            l2, mx = oracle.err_metrics(got, ref)
            assert l2 <= 1e-6 and mx <= 1e-6
        # (b) split invariance + checksum
        ck1 = y.double().sum().item()
        ab1 = y.double().abs().sum().item()
        y2 = torch.empty_like(y)
        f.reset()
        cut = (n // 2) + 12345   # odd: second call starts off-phase and misaligned
        m1 = f.process_device(x.data_ptr(), y2.data_ptr(), cut)
        m2 = f.process_device(x.data_ptr() + 8 * cut, y2.data_ptr() + 8 * m1, n - cut) if cut % 2 == 0 else None
        if m2 is None:
            # device pointers must stay 16-byte aligned: move the tail to an aligned buffer
            tail = x[2 * cut:].clone()
            torch.cuda.synchronize()
            ytail = torch.empty(2 * (m - m1), dtype=torch.float32, device="cuda")
            m2 = f.process_device(tail.data_ptr(), ytail.data_ptr(), n - cut)
            f.synchronize()
            y2[2 * m1:] = ytail
        f.synchronize()
        assert m1 + m2 == m
        assert torch.equal(y, y2)
        assert y2.double().sum().item() == ck1 and y2.double().abs().sum().item() == ab1 and ab1 > 0


@pytest.mark.parametrize("t,d,log2n", [(255, 4, 28), (1023, 1, 28), (127, 1, 26),
                                       (255, 2, 28), (255, 8, 28), (511, 16, 28), (1023, 32, 28), (255, 64, 28),
                                       (255, 12, 28), (255, 24, 28)])
def test_full_size_fft_backend(fir, oracle, gpu_ok, t, d, log2n):
    """Overlap-save backend (AUTO) at BASELINE sizes: windows of the stream within SPEC tolerance of the float64 oracle,
    and split invariance (one call vs two calls at an odd cut) within tolerance of each other.  Decimations 2, 8, 16, 32
    and 64 are the decimating tails of round 3 (frequency-domain alias fold + small inverse) at the same size."""
    import torch
    torch.cuda.set_device(0)
    n = 1 << log2n
    taps = fir.bpf_design(t)
    with fir.IfFir(taps, d, 0) as f:
        assert f.get_backend() == fir.BACKEND_HIP_FFT
        x = torch.empty(2 * n, dtype=torch.float32, device="cuda")
        m = f.out_count(n)
        y = torch.empty(2 * m, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        f.synth_device(x.data_ptr(), 0, n, 0)
        assert f.process_device(x.data_ptr(), y.data_ptr(), n) == m
        f.synchronize()
        w = 8192
        for start in [0, 3840 * 7 - 100, (n // 3) & ~3, n - w]:
            lo = max(0, start - (t - 1))
            xs = x[2 * lo:2 * (start + w)].cpu().numpy()
            hist = np.zeros(2 * (t - 1), dtype=np.float32)
            hist[2 * (t - 1 - (start - lo)):] = xs[:2 * (start - lo)]
            ref = oracle.fir_f64(taps, xs[2 * (start - lo):], d, hist, start)
            first_out = (start + d - 1) // d
            got = y[2 * first_out:2 * first_out + ref.size].cpu().numpy()
            l2, mx = oracle.err_metrics(got, ref)
            assert l2 <= 1e-6 and mx <= 1e-6, (start, l2, mx)
        # completeness: EVERY output of the launch against the direct form (or the one-output-per-thread kernel) on the
        # same input -- windows alone can miss blocks a work-queue bug leaves unwritten
        f.reset()
        f.set_backend(fir.BACKEND_HIP_DIRECT if (t, d) in ((255, 4), (127, 1)) else fir.BACKEND_HIP_GENERIC)
        yd = torch.empty_like(y)
        assert f.process_device(x.data_ptr(), yd.data_ptr(), n) == m
        f.synchronize()
        scale = yd.abs().max().item()
        assert scale > 0.1
        assert (y - yd).abs().max().item() <= 2e-6 * scale   # each side is within 1e-6 of the float64 result
        del yd
        f.set_backend(fir.BACKEND_HIP_FFT)
        f.reset()
        cut = (n // 2) + 12345
        y2 = torch.empty_like(y)
        m1 = f.process_device(x.data_ptr(), y2.data_ptr(), cut)
        tail = x[2 * cut:].clone()
        ytail = torch.empty(2 * (m - m1), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        m2 = f.process_device(tail.data_ptr(), ytail.data_ptr(), n - cut)
        f.synchronize()
        assert m1 + m2 == m
        y2[2 * m1:] = ytail
        scale = y.abs().max().item()
        assert (y - y2).abs().max().item() <= 1e-6 * scale   # block placement differs, values agree to tolerance


def test_full_size_nco_and_index_wrap(fir, oracle, gpu_ok):
    """NCO at BASELINE size (SPEC §3.2): every output against the generic kernel, windows against the float64 oracle;
    then the stream is continued past 2^32 samples (17 calls of 2^28) so that the 32-bit phase arithmetic of the
    kernels wraps: a window of the last call against the oracle at consumed = 16 * 2^28."""
    import torch
    torch.cuda.set_device(0)
    t, d, n = 255, 4, 1 << 28
    taps = fir.bpf_design(t, 0.0, 0.06)
    f_nco = 0.1871
    pw = oracle.nco_phase_word(f_nco)
    with fir.IfFir(taps, d, 0) as f:
        f.set_nco(f_nco)
        assert f.get_backend() == fir.BACKEND_HIP_FFT
        x = torch.empty(2 * n, dtype=torch.float32, device="cuda")
        m = f.out_count(n)
        y = torch.empty(2 * m, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        f.synth_device(x.data_ptr(), 0, n, 3)

        def window_check(start, consumed, w=8192):
            lo = max(0, start - (t - 1))
            xs = x[2 * lo:2 * (start + w)].cpu().numpy()
            if consumed == 0:
                hist = np.zeros(2 * (t - 1), dtype=np.float32)
                hist[2 * (t - 1 - (start - lo)):] = xs[:2 * (start - lo)]
            else:   # the call continues a stream of identical buffers: the history is the tail of x
                assert start == 0
                hist = x[2 * (n - (t - 1)):].cpu().numpy()
            ref = oracle.fir_nco_f64(taps, xs[2 * (start - lo):], d, pw, hist=hist, consumed=consumed + start)
            first_out = (start + d - 1) // d
            got = y[2 * first_out:2 * first_out + ref.size].cpu().numpy()
            l2, mx = oracle.err_metrics(got, ref)
            assert l2 <= 1e-6 and mx <= 1e-6, (start, consumed, l2, mx)

        assert f.process_device(x.data_ptr(), y.data_ptr(), n) == m
        f.synchronize()
        for start in (0, 3840 * 5 - 64, (n // 3) & ~3, n - 8192):
            window_check(start, 0)
        f.reset()
        f.set_backend(fir.BACKEND_HIP_GENERIC)
        yg = torch.empty_like(y)
        f.process_device(x.data_ptr(), yg.data_ptr(), n)
        f.synchronize()
        scale = yg.abs().max().item()
        assert scale > 0.05 and (y - yg).abs().max().item() <= 2e-6 * scale
        del yg
        f.set_backend(fir.BACKEND_HIP_FFT)
        f.reset()
        for _ in range(17):
            f.process_device(x.data_ptr(), y.data_ptr(), n)
        f.synchronize()
        window_check(0, 16 * n)          # absolute indices 2^32 .. 2^32 + 8191


def test_beyond_32_bit_offsets(fir, oracle, gpu_ok):
    """One call of 2^31 + 12 293 samples (17 GB in, 4.3 GB out): sample indices above 2^31 and byte offsets above 2^32
    in every backend's address arithmetic.  Overlap-save against the direct form on every output, windows against the
    float64 oracle at the start, around the 2^32-byte line of the input and of the output, and at the very end."""
    import torch
    torch.cuda.set_device(0)
    t, d = 255, 4
    n = (1 << 31) + 12_293
    taps = fir.bpf_design(t)
    with fir.IfFir(taps, d, 0) as f:
        x = torch.empty(2 * n, dtype=torch.float32, device="cuda")
        m = f.out_count(n)
        y = torch.empty(2 * m, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        f.synth_device(x.data_ptr(), 0, n, 5)
        assert f.get_backend() == fir.BACKEND_HIP_FFT
        assert f.process_device(x.data_ptr(), y.data_ptr(), n) == m
        f.synchronize()
        w = 4096
        for start in (0, (1 << 29) - 2048, (1 << 31) - 2048, n - w):      # 2^29 samples = 2^32 input bytes; 2^31 samples = 2^32 output bytes
            start -= start % d
            lo = start - (t - 1) if start else 0
            xs = x[2 * lo:2 * (start + w)].cpu().numpy()
            assert np.array_equal(xs, oracle.synth_iq(start + w - lo, 5, lo))
            hist = np.zeros(2 * (t - 1), dtype=np.float32)
            hist[2 * (t - 1 - (start - lo)):] = xs[:2 * (start - lo)]
            ref = oracle.fir_f64(taps, xs[2 * (start - lo):], d, hist, start)
            got = y[2 * (start // d):2 * (start // d) + ref.size].cpu().numpy()
            l2, mx = oracle.err_metrics(got, ref)
            assert l2 <= 1e-6 and mx <= 1e-6, (start, l2, mx)
        # decimation 12 = the same tail keeping every 3rd output (round 3): the very bits of every 3rd decimate-by-4 output, with
        # block indices up to 2^31 / 3840 in its index arithmetic
        with fir.IfFir(taps, 12, 0) as f12:
            m12 = f12.out_count(n)
            y12 = torch.empty(2 * m12, dtype=torch.float32, device="cuda")
            torch.cuda.synchronize()
            assert f12.process_device(x.data_ptr(), y12.data_ptr(), n) == m12
            f12.synchronize()
            yv, y12v = y.view(-1, 2), y12.view(-1, 2)
            step12 = 1 << 26
            for a in range(0, m12, step12):
                b = min(m12, a + step12)
                assert torch.equal(y12v[a:b], yv[3 * a:3 * b:3]), a
            del y12, y12v
        f.reset()
        f.set_backend(fir.BACKEND_HIP_DIRECT)
        yd = torch.empty_like(y)
        assert f.process_device(x.data_ptr(), yd.data_ptr(), n) == m
        f.synchronize()
        scale = yd.abs().max().item()
        step = 1 << 28
        worst = max((y[a:a + step] - yd[a:a + step]).abs().max().item() for a in range(0, y.numel(), step))
        assert worst <= 2e-6 * scale, worst / scale


@pytest.mark.parametrize("t,d", [(4095, 1), (4095, 4)])
def test_full_size_two_partition_filter(fir, oracle, gpu_ok, t, d):
    """3074..4096 taps (two partitions, accumulating second launch) on 2^25 samples: every output against the tap-split
    kernel (another algorithm: time-domain MACs), windows against the float64 oracle, the unit impulse returns the taps,
    and a split at a multiple of the block advance is bit-identical to the unsplit call (history = 4096 samples)."""
    import torch
    torch.cuda.set_device(0)
    n = 1 << 25
    taps = fir.bpf_design(t)
    with fir.IfFir(taps, d, 0) as f:
        assert f.get_backend() == fir.BACKEND_HIP_FFT
        x = torch.empty(2 * n, dtype=torch.float32, device="cuda")
        m = f.out_count(n)
        y = torch.empty(2 * m, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        f.synth_device(x.data_ptr(), 0, n, 3)
        assert f.process_device(x.data_ptr(), y.data_ptr(), n) == m
        f.synchronize()
        scale = float(y.abs().max().item())
        for start in [0, 2048 - 16, 4096, (n // 3) & ~3, n - 8192]:
            w = 8192
            lo = max(0, start - (t - 1))
            xs = x[2 * lo:2 * (start + w)].cpu().numpy()
            hist = np.zeros(2 * (t - 1), dtype=np.float32)
            hist[2 * (t - 1 - (start - lo)):] = xs[:2 * (start - lo)]
            ref = oracle.fir_f64(taps, xs[2 * (start - lo):], d, hist, start)
            first_out = (start + d - 1) // d
            got = y[2 * first_out:2 * first_out + ref.size].cpu().numpy()
            l2, mx = oracle.err_metrics(got, ref)
            assert l2 <= 1e-6 and mx <= 1e-6, (start, l2, mx)
        # split at 300 block advances of the 32-row kernel (2048 samples each): same blocks, same bits
        cut = 300 * 2048 * 8
        y2 = torch.empty_like(y)
        f.reset()
        m1 = f.process_device(x.data_ptr(), y2.data_ptr(), cut)
        m2 = f.process_device(x.data_ptr() + 8 * cut, y2.data_ptr() + 8 * m1, n - cut)
        f.synchronize()
        assert m1 + m2 == m and torch.equal(y, y2)
        # another algorithm on the whole array
        yt = torch.empty_like(y)
        f.set_backend(fir.BACKEND_HIP_TAPSPLIT)
        f.reset()
        assert f.process_device(x.data_ptr(), yt.data_ptr(), n) == m
        f.synchronize()
        assert float((y - yt).abs().max().item()) <= 2e-6 * scale
        # impulse response
        f.set_backend(fir.BACKEND_HIP_FFT)
        f.reset()
        x[:2 * 16384].zero_()
        x[2 * 5] = 1.0
        f.process_device(x.data_ptr(), y.data_ptr(), 16384)
        f.synchronize()
        got = y[:2 * (16384 // d)].cpu().numpy()
        full = np.zeros(16384, dtype=np.float64)
        full[5:5 + t] = taps
        want = full[::d][:got.size // 2]
        assert np.max(np.abs(got[0::2] - want)) <= 1e-6 * np.max(np.abs(taps)) and np.max(np.abs(got[1::2])) <= 1e-6 * np.max(np.abs(taps))


@pytest.mark.parametrize("n", [(1 << 23) + 777, (1 << 24) + 12_345, (1 << 25) + 1, 16_773_120])
def test_mid_size_launches_with_the_tail_phase(fir, oracle, gpu_ok, n):
    """Launches of 1 to 6 two-wave rounds hand the remainder out one block per SIMD (the block queue's tail phase, round 3;
    16 773 120 samples = the multi-channel front's chunk): EVERY output against the direct-form kernel on the same input,
    a window at the very end (the tail blocks) against the float64 oracle, and the launch without the tail phase
    (development variant) bit for bit."""
    import torch
    torch.cuda.set_device(0)
    t, d = 255, 4
    taps = fir.bpf_design(t)
    old = os.environ.get("IF_FIR_DEBUG")
    os.environ["IF_FIR_DEBUG"] = "1"
    try:
        with fir.IfFir(taps, d, 0, dev=True) as f:
            x = torch.empty(2 * n, dtype=torch.float32, device="cuda")
            m = f.out_count(n)
            y = torch.empty(2 * m, dtype=torch.float32, device="cuda")
            torch.cuda.synchronize()
            f.synth_device(x.data_ptr(), 0, n, 3)
            assert f.get_backend() == fir.BACKEND_HIP_FFT
            assert f.process_device(x.data_ptr(), y.data_ptr(), n) == m
            f.synchronize()
            assert f.debug_queue_faults() == 0
            w = 8192
            start = (n - w) & ~3
            xs = x[2 * (start - (t - 1)):].cpu().numpy()
            hist = xs[:2 * (t - 1)].copy()
            ref = oracle.fir_f64(taps, xs[2 * (t - 1):], d, hist, start)
            got = y[2 * (start // d):].cpu().numpy()
            l2, mx = oracle.err_metrics(got, ref)
            assert l2 <= 1e-6 and mx <= 1e-6, (l2, mx)
            f.reset()
            f.set_tuning(1000000 + 256)          # the same launch without the tail phase: the same bits
            y2 = torch.empty_like(y)
            assert f.process_device(x.data_ptr(), y2.data_ptr(), n) == m
            f.synchronize()
            assert torch.equal(y, y2)
            f.set_tuning(0)
            f.reset()
            f.set_backend(fir.BACKEND_HIP_DIRECT)
            assert f.process_device(x.data_ptr(), y2.data_ptr(), n) == m
            f.synchronize()
            scale = y2.abs().max().item()
            assert scale > 0.1 and (y - y2).abs().max().item() <= 2e-6 * scale
    finally:
        if old is None:
            os.environ.pop("IF_FIR_DEBUG", None)
        else:
            os.environ["IF_FIR_DEBUG"] = old


@pytest.mark.parametrize("d,nch", [(8, 16), (8, 6), (16, 16)])
def test_full_size_filter_bank(fir, oracle, gpu_ok, monkeypatch, d, nch):
    """The filter bank at the headline size (2^28 samples of one wideband stream): EVERY output of every channel against (a) the
    other form of the same bank -- decimation 8: the all-slots launches (round 4) against the per-channel form (development launch
    4096) -- and (b), for two channels, a context that mixes, filters and decimates that one channel (if_fir_set_nco); windows of one
    channel against the float64 NCO oracle; one call = two calls at an odd cut."""
    import torch
    torch.cuda.set_device(0)
    monkeypatch.setenv("IF_FIR_DEBUG", "1")
    n, t = 1 << 28, 255
    taps = fir.bpf_design(t, 0.0, 0.02)
    slots = list(range(16)) if nch == 16 else [1, 3, 5, 7, 9, 2]   # (6: five odd slots through the all-slots launch + an even one per channel)
    with fir.IfFir(taps, d, 0, dev=True) as f:
        x = torch.empty(2 * n, dtype=torch.float32, device="cuda")
        m = f.out_count(n)
        outs = [torch.empty(2 * m, dtype=torch.float32, device="cuda") for _ in slots]
        torch.cuda.synchronize()
        f.synth_device(x.data_ptr(), 0, n, 5)
        assert f.get_backend() == fir.BACKEND_HIP_FFT
        assert f.channelizer_process_device(slots, x.data_ptr(), [o.data_ptr() for o in outs], n) == m
        f.synchronize()
        assert f.debug_queue_faults() == 0
        scale = max(o.abs().max().item() for o in outs)
        assert scale > 0.05
        if d == 8:
            # (a) the per-channel form, every output of every channel
            f.set_tuning(1000000 + 4096)
            f.reset()
            other = torch.empty(2 * m, dtype=torch.float32, device="cuda")
            for c, s in enumerate(slots):
                assert f.channelizer_process_device([s], x.data_ptr(), [other.data_ptr()], n) == m
                f.synchronize()
                f.reset()
                assert (outs[c] - other).abs().max().item() <= 2e-6 * scale, (d, s)
            del other
            f.set_tuning(0)
        # split invariance: two calls at an odd cut (the all-slots launches carry the history and the mix-down phase like any call)
        f.reset()
        cut = (n // 2) + 12345
        part = [torch.empty(2 * m, dtype=torch.float32, device="cuda") for _ in slots]
        m1 = f.channelizer_process_device(slots, x.data_ptr(), [p.data_ptr() for p in part], cut)
        tail = x[2 * cut:].clone()
        torch.cuda.synchronize()
        m2 = f.channelizer_process_device(slots, tail.data_ptr(), [p.data_ptr() + 8 * m1 for p in part], n - cut)
        f.synchronize()
        assert m1 + m2 == m
        for c in range(len(slots)):
            assert (outs[c] - part[c]).abs().max().item() <= 1e-6 * scale, (d, slots[c])
        del part, tail
    # (b) one context per channel, every output
    ref = torch.empty(2 * m, dtype=torch.float32, device="cuda")
    for c in (1, len(slots) - 1):
        s = slots[c]
        with fir.IfFir(taps, d, 0) as f1:
            f1.set_nco(s / 16.0 if s <= 8 else s / 16.0 - 1.0)
            assert f1.process_device(x.data_ptr(), ref.data_ptr(), n) == m
            f1.synchronize()
        assert (outs[c] - ref).abs().max().item() <= 2e-6 * scale, (d, s)
    # windows of one channel against the float64 NCO oracle
    c = 2
    s = slots[c]
    w = 8192
    for start in [0, (n // 3) & ~15, n - w]:
        lo = max(0, start - (t - 1))
        xs = x[2 * lo:2 * (start + w)].cpu().numpy()
        hist = np.zeros(2 * (t - 1), dtype=np.float32)
        hist[2 * (t - 1 - (start - lo)):] = xs[:2 * (start - lo)]
        refw = oracle.fir_nco_f64(taps, xs[2 * (start - lo):], d, (s << 28) & 0xFFFFFFFF, hist, start)
        first_out = (start + d - 1) // d
        got = outs[c][2 * first_out:2 * first_out + refw.size].cpu().numpy()
        l2, mx = oracle.err_metrics(got, refw)
        assert l2 <= 1e-6 and mx <= 1e-6, (d, s, start, l2, mx)
