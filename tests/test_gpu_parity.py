"""GPU parity tests (run with -m gpu on an MI355X): every call goes through the C-ABI of libif_fir.so and is compared
with the oracle — bit-for-bit against the float32 fma-order model and within SPEC §3 tolerance of the float64 oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "if_fir_golden.npz")
SEG = dict(seg_mode=1, seg_len=32)
TOL = 1e-6  # SPEC §3 / north_star: 1e-6 relative (norm-wise and max-wise)


@pytest.fixture(scope="module")
def torch_cuda(gpu_ok):
    import torch
    torch.cuda.set_device(0)
    return torch


def _check(oracle, y, taps, x, d, hist=None, consumed=0):
    model = oracle.fir_f32fma(taps, x, d, hist, consumed, **SEG)
    assert y.shape == model.shape
    assert np.array_equal(y, model), "max |diff| vs order model = %g" % np.max(np.abs(y - model))
    ref = oracle.fir_f64(taps, x, d, hist, consumed)
    if ref.size and np.any(ref):
        l2, mx = oracle.err_metrics(y, ref)
        assert l2 <= TOL and mx <= TOL, (l2, mx)


def test_library_loaded_is_in_tree(fir, gpu_ok):
    assert os.path.dirname(fir.LIB_PATH).endswith("qo-100-tools_amd") and fir.lib() is not None


def test_auto_backend_policy(fir, gpu_ok):
    """AUTO = overlap-save for every filter the library accepts (any decimation: it is the fastest in every case
    measured, tools/policy_sweep.py); 3074..4096 taps run as two partitions."""
    expect = {(255, 4): fir.BACKEND_HIP_FFT, (255, 1): fir.BACKEND_HIP_FFT, (127, 1): fir.BACKEND_HIP_FFT,
              (1023, 1): fir.BACKEND_HIP_FFT, (1023, 4): fir.BACKEND_HIP_FFT, (31, 1): fir.BACKEND_HIP_FFT,
              (3, 1): fir.BACKEND_HIP_FFT, (255, 2): fir.BACKEND_HIP_FFT, (1023, 8): fir.BACKEND_HIP_FFT,
              (255, 16): fir.BACKEND_HIP_FFT, (63, 64): fir.BACKEND_HIP_FFT, (2047, 1): fir.BACKEND_HIP_FFT,
              (3073, 4): fir.BACKEND_HIP_FFT, (3075, 1): fir.BACKEND_HIP_FFT, (4095, 4): fir.BACKEND_HIP_FFT,
              (4095, 64): fir.BACKEND_HIP_FFT}
    for (t, d), b in expect.items():
        with fir.IfFir(fir.bpf_design(t), d, 16) as f:
            assert f.get_backend() == b, (t, d)


def test_synth_device_bit_exact(fir, oracle, torch_cuda):
    torch = torch_cuda
    with fir.IfFir(fir.bpf_design(127), 1, 1 << 16) as f:
        for first, n, ch in [(0, 10007, 0), (123456789012, 4096, 7), (5, 1, 1), (2 ** 40 + 3, 70001, 255)]:
            buf = torch.empty(2 * n, dtype=torch.float32, device="cuda")
            f.synth_device(buf.data_ptr(), first, n, ch)
            f.synchronize()
            assert np.array_equal(buf.cpu().numpy(), oracle.synth_iq(n, ch, first))


@pytest.mark.parametrize("t,d", [(255, 4), (255, 1), (127, 1), (127, 4)])
@pytest.mark.parametrize("variant", list(range(7)))
def test_direct_kernels_vs_oracle(fir, oracle, t, d, variant):
    taps = fir.bpf_design(t)
    rng = np.random.default_rng(100 * t + d)
    n = 70_001
    x = np.concatenate([oracle.synth_iq(n // 2), rng.standard_normal(2 * (n - n // 2)).astype(np.float32)])
    with fir.IfFir(taps, d, n, backend=fir.BACKEND_HIP_DIRECT) as f:
        assert f.get_backend() == fir.BACKEND_HIP_DIRECT
        f.set_tuning(variant)
        _check(oracle, f.process(x), taps, x, d)


@pytest.mark.parametrize("t,d", [(1, 1), (2, 2), (31, 3), (64, 1), (255, 5), (1023, 1), (1023, 4), (4096, 64)])
def test_generic_kernel_vs_oracle(fir, oracle, t, d):
    rng = np.random.default_rng(t + d)
    taps = (fir.bpf_design(t) if t % 2 and t >= 3 else rng.standard_normal(t).astype(np.float32) / t)
    n = 20_011
    x = rng.standard_normal(2 * n).astype(np.float32)
    with fir.IfFir(taps, d, n, backend=fir.BACKEND_HIP_GENERIC) as f:
        assert f.get_backend() == fir.BACKEND_HIP_GENERIC
        _check(oracle, f.process(x), taps, x, d)


def test_generic_backend_forced_equals_direct(fir, oracle):
    taps = fir.bpf_design(255)
    x = oracle.synth_iq(30_000, 2)
    with fir.IfFir(taps, 4, 30_000, backend=fir.BACKEND_HIP_DIRECT) as f:
        a = f.process(x)
        f.reset()
        f.set_backend(fir.BACKEND_HIP_GENERIC)
        b = f.process(x)
        assert np.array_equal(a, b)
        with pytest.raises(fir.IfFirError):
            f.set_backend(9)                          # unknown backend: refused with a message, context stays usable
        f.reset()
        assert np.array_equal(f.process(x), b)
        f.set_backend(fir.BACKEND_AUTO)          # AUTO = overlap-save here: same answer within SPEC tolerance
        assert f.get_backend() == fir.BACKEND_HIP_FFT
        f.reset()
        l2, mx = oracle.err_metrics(f.process(x), oracle.fir_f64(taps, x, 4))
        assert l2 <= TOL and mx <= TOL


@pytest.mark.parametrize("t", [127, 255, 1023])
@pytest.mark.parametrize("d", [1, 4])
def test_golden_vectors(fir, oracle, t, d):
    g = np.load(GOLD)
    taps, x, ref = g["taps_%d" % t], g["x"], g["y_T%d_D%d" % (t, d)]
    with fir.IfFir(taps, d, 4096) as f:
        y = f.process(x)
    l2, mx = oracle.err_metrics(y, ref)
    assert l2 <= TOL and mx <= TOL, (l2, mx)


def test_config0_real_samples_through_complex_path(fir, oracle):
    """BASELINE configs[0]: 127 taps over 2^20 real float samples (I = x, Q = 0)."""
    g = np.load(GOLD)
    taps = g["taps_127"]
    xr = oracle.synth_iq(1 << 20)[0::2].copy()
    x = np.zeros(2 * xr.size, dtype=np.float32)
    x[0::2] = xr
    ref = oracle.fir_real_f64(taps, xr)
    with fir.IfFir(taps, 1, xr.size, backend=fir.BACKEND_HIP_DIRECT) as f:
        y = f.process(x)
        l2, mx = oracle.err_metrics(y[0::2], ref)
        assert l2 <= TOL and mx <= TOL and not np.any(y[1::2])      # direct form: Q stays exactly zero
        f.set_backend(fir.BACKEND_AUTO)                              # overlap-save: Q is rounding noise
        f.reset()
        y = f.process(x)
    l2, mx = oracle.err_metrics(y[0::2], ref)
    assert l2 <= TOL and mx <= TOL and np.max(np.abs(y[1::2])) <= TOL * np.max(np.abs(ref))
    # and the committed 4096-sample real fixture
    x4 = np.zeros(2 * 4096, dtype=np.float32)
    x4[0::2] = g["xr"]
    with fir.IfFir(taps, 1, 4096) as f:
        l2, mx = oracle.err_metrics(f.process(x4)[0::2], g["yr_T127"])
    assert l2 <= TOL and mx <= TOL


@pytest.mark.parametrize("t,d", [(255, 4), (255, 1), (127, 1), (31, 3)])
def test_streaming_ragged_pieces(fir, oracle, t, d):
    """process(a‖b‖…) ≡ process(a), process(b), …  — history and decimation phase carried, including empty pieces,
    pieces shorter than the history, odd lengths (misaligned phase) and tile-boundary lengths."""
    taps = fir.bpf_design(t)
    n = 40_000
    x = oracle.synth_iq(n, 1)
    one = oracle.fir_f32fma(taps, x, d, **SEG)
    cuts = [0, 0, 1, 3, 10, 11, 200, 253, 254, 255, 511, 8192 + 511, 8192 + 512, 2 * 8192 + 513, 30_001, n]
    bit_exact_backend = fir.BACKEND_HIP_DIRECT if (t, d) != (31, 3) else fir.BACKEND_HIP_GENERIC
    with fir.IfFir(taps, d, n, backend=bit_exact_backend) as f:
        parts = [f.process(x[2 * a:2 * b]) for a, b in zip(cuts[:-1], cuts[1:])]
        assert np.array_equal(np.concatenate(parts), one)
        f.reset()
        assert np.array_equal(f.process(x), one)   # reset restores zero history and phase 0


@pytest.mark.parametrize("n", [1, 2, 3, 4, 7, 254, 255, 256, 2047, 2048, 2049, 8191, 8192, 8193, 16384, 24577])
def test_sizes_around_tile_edges(fir, oracle, n):
    for t, d in [(255, 4), (255, 1)]:
        taps = fir.bpf_design(t)
        x = oracle.synth_iq(n, 3)
        with fir.IfFir(taps, d, n, backend=fir.BACKEND_HIP_DIRECT) as f:
            _check(oracle, f.process(x), taps, x, d)


def test_impulse_and_zero_input(fir, oracle):
    taps = fir.bpf_design(255)
    x = np.zeros(2 * 1000, dtype=np.float32)
    x[0], x[1] = 1.0, -2.0
    with fir.IfFir(taps, 1, 1000, backend=fir.BACKEND_HIP_DIRECT) as f:
        y = f.process(x)
        assert np.array_equal(y[0:510:2], taps) and np.array_equal(y[1:510:2], -2 * taps) and not np.any(y[510:])
        assert f.process(np.zeros(0, dtype=np.float32)).size == 0
        assert not np.any(f.process(np.zeros(2 * 600, dtype=np.float32))[2 * 255:])


def test_device_api_canaries_and_errors(fir, oracle, torch_cuda):
    """process_device on raw pointers: outputs land exactly in [0, M) (guard words untouched), misaligned pointers
    are refused with a message, and the context keeps working afterwards."""
    torch = torch_cuda
    taps = fir.bpf_design(255)
    n, d = 50_001, 4
    x = oracle.synth_iq(n, 4)
    m = oracle.out_count(0, n, d)
    guard = 1024
    xin = torch.from_numpy(x).cuda()
    out = torch.full((2 * m + 2 * guard,), 12345.0, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    with fir.IfFir(taps, d, n, backend=fir.BACKEND_HIP_DIRECT, dev=True) as f:
        with pytest.raises(fir.IfFirError, match="aligned"):
            f.process_device(xin.data_ptr() + 4, out.data_ptr(), n - 1)
        got = f.process_device(xin.data_ptr(), out.data_ptr() + 4 * guard, n)
        f.synchronize()
        assert got == m
        o = out.cpu().numpy()
        assert np.all(o[:guard] == 12345.0) and np.all(o[guard + 2 * m:] == 12345.0)
        assert np.array_equal(o[guard:guard + 2 * m], oracle.fir_f32fma(taps, x, d, **SEG))
        ms = f.time_device(xin.data_ptr(), out.data_ptr(), n, 1, 3)   # timing helper leaves the stream state alone
        assert ms > 0
        again = f.process_device(xin.data_ptr(), out.data_ptr(), n)
        f.synchronize()
        st = oracle.OracleStream(taps, d)
        st.process(x, "f32", **SEG)
        assert np.array_equal(out.cpu().numpy()[:2 * again], st.process(x, "f32", **SEG))
    with pytest.raises(fir.IfFirError):
        fir.IfFir(taps, 4, n, device=99)


def test_caller_stream(fir, oracle, torch_cuda):
    torch = torch_cuda
    taps = fir.bpf_design(127)
    n = 20_000
    x = oracle.synth_iq(n)
    s = torch.cuda.Stream()
    with fir.IfFir(taps, 1, n, backend=fir.BACKEND_HIP_DIRECT) as f, torch.cuda.stream(s):
        f.set_stream(s.cuda_stream)
        xin = torch.from_numpy(x).cuda(non_blocking=False)
        out = torch.empty(2 * n, dtype=torch.float32, device="cuda")
        f.process_device(xin.data_ptr(), out.data_ptr(), n)
        s.synchronize()
        assert np.array_equal(out.cpu().numpy(), oracle.fir_f32fma(taps, x, 1, **SEG))


@pytest.mark.parametrize("t", [1023, 255, 127, 1025, 257, 31, 1, 259, 513, 515])
def test_fft_backend_vs_oracle(fir, oracle, t):
    """Overlap-save FFT-FIR: no bit-exact model, SPEC §3 tolerance against the float64 oracle; compared
    with the direct/generic form on the same input."""
    rng = np.random.default_rng(t)
    taps = fir.bpf_design(t) if (t % 2 and t >= 3) else np.array([0.75], dtype=np.float32)
    n = 50_021
    x = np.concatenate([oracle.synth_iq(n // 2, 5), rng.standard_normal(2 * (n - n // 2)).astype(np.float32)])
    ref = oracle.fir_f64(taps, x, 1)
    with fir.IfFir(taps, 1, n, backend=fir.BACKEND_HIP_DIRECT if t in (127, 255) else fir.BACKEND_HIP_GENERIC) as f:
        direct = f.process(x)
        f.reset()
        f.set_backend(fir.BACKEND_HIP_FFT)
        assert f.get_backend() == fir.BACKEND_HIP_FFT
        y = f.process(x)
        l2, mx = oracle.err_metrics(y, ref)
        assert l2 <= TOL and mx <= TOL, (l2, mx)
        l2d, mxd = oracle.err_metrics(direct, ref)
        assert l2d <= TOL and mxd <= TOL
        # streaming in ragged pieces: history carried across calls through the FFT path too
        f.reset()
        cuts = [0, 1, 500, 1022, 1023, 4096, 4097, 3840 * 3 + 5, 30_000, n]
        parts = [f.process(x[2 * a:2 * b]) for a, b in zip(cuts[:-1], cuts[1:])]
        l2, mx = oracle.err_metrics(np.concatenate(parts), ref)
        assert l2 <= TOL and mx <= TOL, (l2, mx)


@pytest.mark.parametrize("n", [1, 63, 64, 3071, 3072, 3073, 3839, 3840, 3841, 4096, 8191, 12289])
def test_fft_backend_block_edges(fir, oracle, n):
    for t in (1023, 255):
        taps = fir.bpf_design(t)
        x = oracle.synth_iq(n, 6)
        with fir.IfFir(taps, 1, n) as f:
            f.set_backend(fir.BACKEND_HIP_FFT)
            y = f.process(x)
        ref = oracle.fir_f64(taps, x, 1)
        scale = max(np.max(np.abs(ref)), 1e-3)
        assert np.max(np.abs(y - ref)) <= 1e-6 * max(scale, 0.5), (t, n, np.max(np.abs(y - ref)))


@pytest.mark.parametrize("t,d,i16", [(255, 4, False), (255, 1, False), (1023, 4, False), (255, 4, True)])
def test_fft_backend_run_queue_on_small_grid(fir, oracle, t, d, i16):
    """The overlap-save kernel hands out runs of blocks through an atomic queue, which a normal-size test input never
    reaches (256 CUs x 8 waves each take one run and are done).  Tuning 2000+k launches at most k workgroups: ~270
    blocks on 8 or 24 waves go through every stage of the guided schedule (runs of RA, runs of RB, single blocks).
    Which wave computes a block must not matter: bit-identical to the default launch, and within SPEC tolerance."""
    n = 1_050_007
    taps = fir.bpf_design(t)
    if i16:
        xi = np.random.default_rng(77).integers(-32768, 32768, 2 * n, dtype=np.int16)
        x = xi.astype(np.float32) * np.float32(2.0 ** -15)
    else:
        x = oracle.synth_iq(n, 21)
    ref = oracle.fir_f64(taps, x, d)
    with fir.IfFir(taps, d, n, dev=True) as f:
        if i16:
            f.set_input_format(fir.INPUT_I16)
        f.set_backend(fir.BACKEND_HIP_FFT)
        src = xi if i16 else x
        y0 = f.process(src)
        l2, mx = oracle.err_metrics(y0, ref)
        assert l2 <= TOL and mx <= TOL, (l2, mx)
        for k in (1, 3):
            f.reset()
            f.set_tuning(2000 + k)
            yk = f.process(src)
            assert np.array_equal(yk, y0), (k, int(np.argmax(yk != y0)))
        # streaming through the limited grid as well: the first block of a call takes the history path
        f.reset()
        cuts = [0, 333_333, 700_001, n]
        parts = [f.process(src[2 * a:2 * b]) for a, b in zip(cuts[:-1], cuts[1:])]
        l2, mx = oracle.err_metrics(np.concatenate(parts), ref)
        assert l2 <= TOL and mx <= TOL, (l2, mx)
        assert f.debug_queue_faults() == 0      # no bounded wait of the block queue expired


@pytest.mark.parametrize("t,d,i16", [(255, 4, False), (127, 1, False), (255, 8, True), (255, 2, False), (255, 5, False)])
def test_fft_backend_single_round_launches(fir, oracle, t, d, i16, monkeypatch):
    """Round 5: a call of at most one block per wave of the chip is a single-round launch -- one block per wave, dealt slot-major over all CUs, no block
    queue.  Which wave of which workgroup computes a block must not matter: calls of 1 ... 300 blocks (and one just beyond the single-round limit of a 2-workgroup grid)
    against the same calls with single-round launches switched off (development tuning 1000000 + 262144), bit for bit, and against the oracle."""
    monkeypatch.setenv("IF_FIR_DEBUG", "1")
    taps = fir.bpf_design(t)
    rng = np.random.default_rng(91)
    for n in (1, 257, 3_900, 19_000, 65_536, 1_150_000):
        if i16:
            xi = rng.integers(-32768, 32768, 2 * n, dtype=np.int16)
            x = xi.astype(np.float32) * np.float32(2.0 ** -15)
        else:
            x = oracle.synth_iq(n, 23 + n % 7)
        src = xi if i16 else x
        ref = oracle.fir_f64(taps, x, d)
        with fir.IfFir(taps, d, n, dev=True) as f:
            if i16:
                f.set_input_format(fir.INPUT_I16)
            f.set_backend(fir.BACKEND_HIP_FFT)
            y0 = f.process(src)
            if n >= 3000:    # (shorter calls end inside the filter's rise: the error metric is relative to the largest output)
                l2, mx = oracle.err_metrics(y0, ref)
                assert l2 <= TOL and mx <= TOL, (n, l2, mx)
            else:
                assert np.allclose(y0, ref, rtol=0.0, atol=2e-6), n
            for tuning in (1262144, 2002):          # single-round launches off; a 2-workgroup grid (single round up to 16 blocks, the queue beyond)
                f.reset()
                f.set_tuning(tuning)
                y1 = f.process(src)
                assert np.array_equal(y1, y0), (n, tuning, int(np.argmax(y1 != y0)) if len(y0) else -1)
            assert f.debug_queue_faults() == 0


def test_diagnostic_variants_need_the_debug_switch(fir, gpu_ok):
    """The product library takes schedule variants 0..6 only; the development library (include/if_fir_debug.h) has the
    grid limits and, with IF_FIR_DEBUG=1, the diagnostic launches that skip loads or stores (wrong results)."""
    old = os.environ.pop("IF_FIR_DEBUG", None)
    try:
        with fir.IfFir(fir.bpf_design(255), 4, 1000) as f:
            for v in (7, 10, 1001, 2003, 4000, 1000008):
                with pytest.raises(fir.IfFirError, match="development"):
                    f.set_tuning(v)
            f.set_tuning(3)
            f.set_tuning(0)
        with fir.IfFir(fir.bpf_design(255), 4, 1000, dev=True) as f:
            with pytest.raises(fir.IfFirError, match="diagnostic"):
                f.set_tuning(1001)
            f.set_tuning(2003)
            f.set_tuning(0)
            os.environ["IF_FIR_DEBUG"] = "1"
            f.set_tuning(1032)
            f.set_tuning(0)
    finally:
        os.environ.pop("IF_FIR_DEBUG", None)
        if old is not None:
            os.environ["IF_FIR_DEBUG"] = old


def test_fft_backend_rejects_unsupported(fir):
    with fir.IfFir(fir.bpf_design(255), 3, 1000) as f:
        f.set_backend(fir.BACKEND_HIP_FFT)            # any decimation (full-rate kernel + selecting store)
    with fir.IfFir(fir.bpf_design(3075), 4, 1000) as f:
        f.set_backend(fir.BACKEND_HIP_FFT)            # more than 3073 taps: two partitions of up to 2048 taps
        out = [0] * 2
        with pytest.raises(fir.IfFirError, match="filter bank|needs <= 3073 taps"):
            f.channelizer_process_device([1, 3], 0, out, 16)   # the filter bank is a single-partition kernel


@pytest.mark.parametrize("t", [255, 127, 1023, 257, 259, 513])
def test_fft_backend_decimate4_vs_oracle(fir, oracle, t):
    """Overlap-save with the decimation folded into the frequency domain (1024-point inverse): SPEC tolerance against
    the float64 oracle, one-shot and in ragged pieces (odd piece lengths exercise every decimation phase n0)."""
    rng = np.random.default_rng(40 + t)
    taps = fir.bpf_design(t)
    n = 60_013
    x = np.concatenate([oracle.synth_iq(n // 2, 9), rng.standard_normal(2 * (n - n // 2)).astype(np.float32)])
    ref = oracle.fir_f64(taps, x, 4)
    with fir.IfFir(taps, 4, n) as f:
        f.set_backend(fir.BACKEND_HIP_FFT)
        y = f.process(x)
        assert y.shape == ref.shape
        l2, mx = oracle.err_metrics(y, ref)
        assert l2 <= TOL and mx <= TOL, (l2, mx)
        f.reset()
        cuts = [0, 1, 2, 3, 7, 500, 1021, 3841, 3840 * 2 + 6, 20_001, 40_003, n]
        parts = [f.process(x[2 * a:2 * b]) for a, b in zip(cuts[:-1], cuts[1:])]
        l2, mx = oracle.err_metrics(np.concatenate(parts), ref)
        assert l2 <= TOL and mx <= TOL, (l2, mx)


@pytest.mark.parametrize("n", [1, 3, 4, 5, 959 * 4, 960 * 4, 960 * 4 + 1, 3841, 4096, 7681, 15361])
def test_fft_backend_decimate4_block_edges(fir, oracle, n):
    taps = fir.bpf_design(255)
    x = oracle.synth_iq(n, 8)
    with fir.IfFir(taps, 4, n) as f:
        f.set_backend(fir.BACKEND_HIP_FFT)
        y = f.process(x)
    ref = oracle.fir_f64(taps, x, 4)
    assert y.shape == ref.shape
    assert np.max(np.abs(y - ref)) <= 1e-6 * max(np.max(np.abs(ref)), 0.5), (n, np.max(np.abs(y - ref)))


TS = dict(seg_mode=3, seg_len=32)


@pytest.mark.parametrize("t,d", [(1, 1), (2, 2), (3, 1), (5, 3), (31, 3), (64, 1), (255, 4), (255, 5), (1023, 1),
                                 (1023, 4), (2047, 8), (4096, 64), (4095, 1), (127, 16)])
def test_tapsplit_kernel_vs_oracle(fir, oracle, t, d):
    """Tap-split kernel (taps in LDS, 4 lanes per output, DPP reduction): bit-exact against the oracle's order model
    (mode 3) and within SPEC tolerance of the float64 oracle, for any (T, D), one-shot and streamed."""
    rng = np.random.default_rng(7 * t + d)
    taps = (fir.bpf_design(t) if t % 2 and t >= 3 else rng.standard_normal(t).astype(np.float32) / max(t, 1))
    n = 30_011
    x = np.concatenate([oracle.synth_iq(n // 2, 11), rng.standard_normal(2 * (n - n // 2)).astype(np.float32)])
    model = oracle.fir_f32fma(taps, x, d, **TS)
    ref = oracle.fir_f64(taps, x, d)
    with fir.IfFir(taps, d, n, backend=fir.BACKEND_HIP_TAPSPLIT) as f:
        assert f.get_backend() == fir.BACKEND_HIP_TAPSPLIT
        y = f.process(x)
        assert np.array_equal(y, model), np.max(np.abs(y - model))
        if np.any(ref):
            l2, mx = oracle.err_metrics(y, ref)
            assert l2 <= TOL and mx <= TOL, (l2, mx)
        f.reset()
        cuts = [0, 1, 2, 77, 255, 256, 4097, 9000, 20_001, n]
        parts = [f.process(x[2 * a:2 * b]) for a, b in zip(cuts[:-1], cuts[1:])]
        assert np.array_equal(np.concatenate(parts), model)


@pytest.mark.parametrize("t,d", [(255, 4), (255, 1), (127, 1), (1023, 4), (31, 3), (5, 1)])
def test_complex_taps(fir, oracle, t, d):
    """Complex taps (channel selection): overlap-save backend within SPEC tolerance of the float64 oracle; generic
    kernel bit-exact against its order model; streaming pieces; the real-tap-only backends refuse."""
    rng = np.random.default_rng(t * 3 + d)
    g = fir.bpf_design_complex(t, 0.2, 0.1) if t >= 31 else rng.standard_normal(2 * t).astype(np.float32)
    n = 40_009
    x = np.concatenate([oracle.synth_iq(n // 2, 12), rng.standard_normal(2 * (n - n // 2)).astype(np.float32)])
    ref = oracle.fir_ctaps_f64(g, x, d)
    with fir.IfFir(g, d, n, complex_taps=True) as f:
        assert f.get_backend() == fir.BACKEND_HIP_FFT       # complex taps: overlap-save for every decimation
        y = f.process(x)
        l2, mx = oracle.err_metrics(y, ref)
        assert l2 <= TOL and mx <= TOL, (l2, mx)
        f.reset()
        cuts = [0, 3, 1000, 1001, 8191, 20_000, n]
        parts = [f.process(x[2 * a:2 * b]) for a, b in zip(cuts[:-1], cuts[1:])]
        l2, mx = oracle.err_metrics(np.concatenate(parts), ref)
        assert l2 <= TOL and mx <= TOL
        f.set_backend(fir.BACKEND_HIP_GENERIC)
        f.reset()
        yg = f.process(x)
        assert np.array_equal(yg, oracle.fir_ctaps_f32fma(g, x, d))
        for b in (fir.BACKEND_HIP_DIRECT, fir.BACKEND_HIP_TAPSPLIT):
            with pytest.raises(fir.IfFirError):
                f.set_backend(b)


def test_complex_taps_golden(fir, oracle):
    gold = np.load(GOLD)
    for d in (1, 4):
        with fir.IfFir(gold["ctaps_255"], d, 4096, complex_taps=True) as f:
            l2, mx = oracle.err_metrics(f.process(gold["x"]), gold["yc_T255_D%d" % d])
        assert l2 <= TOL and mx <= TOL


@pytest.mark.parametrize("t,d", [(255, 4), (255, 1), (1023, 1), (127, 4), (31, 3)])
def test_int16_input_front_end(fir, oracle, t, d):
    """int16 IQ input: value = int16 * 2^-15 converted inside the kernels' loads.  Overlap-save backend
    within SPEC tolerance, generic kernel bit-exact, ragged streaming, and the real-only backends refuse the format."""
    rng = np.random.default_rng(900 + t + d)
    taps = fir.bpf_design(t)
    n = 50_007
    xi = np.clip(np.round(oracle.synth_iq(n, 13) * 16384.0), -32768, 32767).astype(np.int16)
    xi[: 2 * 100] = rng.integers(-32768, 32768, 2 * 100).astype(np.int16)       # full-scale corner values
    xf = xi.astype(np.float32) * np.float32(2.0 ** -15)
    ref = oracle.fir_f64(taps, xf, d)
    with fir.IfFir(taps, d, n) as f:
        f.set_input_format(fir.INPUT_I16)
        assert f.get_backend() == fir.BACKEND_HIP_FFT
        y = f.process(xi)
        l2, mx = oracle.err_metrics(y, ref)
        assert l2 <= TOL and mx <= TOL, (l2, mx)
        f.reset()
        cuts = [0, 1, 255, 3841, 20_000, n]
        parts = [f.process(xi[2 * a:2 * b]) for a, b in zip(cuts[:-1], cuts[1:])]
        l2, mx = oracle.err_metrics(np.concatenate(parts), ref)
        assert l2 <= TOL and mx <= TOL
        f.set_backend(fir.BACKEND_HIP_GENERIC)
        f.reset()
        assert np.array_equal(f.process(xi), oracle.fir_f32fma(taps, xf, d, seg_mode=1, seg_len=32))
        for b in (fir.BACKEND_HIP_DIRECT, fir.BACKEND_HIP_TAPSPLIT):
            with pytest.raises(fir.IfFirError):
                f.set_backend(b)
    with fir.IfFir(taps, d, n, backend=fir.BACKEND_HIP_TAPSPLIT) as f:
        with pytest.raises(fir.IfFirError):
            f.set_input_format(fir.INPUT_I16)        # forced real-only backend: format refused, context usable
        assert f.process(xf).size == 2 * oracle.out_count(0, n, d)


def test_multi_channel_front_one_rank(fir, oracle, torch_cuda):
    """if_fir_mc_* with world = 1: three channels with their own taps and their own streaming state on one GPU, device
    pointer arrays in and out, two calls (history carried per channel); no RCCL is loaded on this path."""
    torch = torch_cuda
    n, d, t = 70_001, 4, 255
    bands = [(0.15, 0.25), (0.02, 0.08), (0.30, 0.45)]
    taps = np.stack([fir.bpf_design(t, lo, hi) for lo, hi in bands])
    xs = [oracle.synth_iq(n, 30 + c) for c in range(3)]
    dev_in = [torch.from_numpy(x).cuda() for x in xs]
    cuts = [0, 30_003, n]
    with fir.IfFirMc(taps, d, n) as mc:
        assert [fir.mc_owner(c, 1) for c in range(3)] == [0, 0, 0]
        assert all(mc.channel_ctx(c) for c in range(3)) and mc.channel_ctx(3) is None
        for backend, exact in ((fir.BACKEND_HIP_DIRECT, True), (fir.BACKEND_AUTO, False)):
            mc.set_backend(backend)
            mc.reset()
            parts = [[] for _ in range(3)]
            for a, b in zip(cuts[:-1], cuts[1:]):
                m_exp = oracle.out_count(a, b - a, d)
                outs = [torch.full((2 * m_exp + 8,), 7.0, dtype=torch.float32, device="cuda") for _ in range(3)]
                tails = [x[2 * a:2 * b].clone() for x in dev_in]       # 16-byte aligned pieces
                torch.cuda.synchronize()
                m = mc.process_device([p.data_ptr() for p in tails], [o.data_ptr() for o in outs], b - a)
                assert m == m_exp
                for c in range(3):
                    o = outs[c].cpu().numpy()
                    assert np.all(o[2 * m:] == 7.0)                    # nothing written past the outputs
                    parts[c].append(o[:2 * m])
            for c in range(3):
                y = np.concatenate(parts[c])
                if exact:
                    assert np.array_equal(y, oracle.fir_f32fma(taps[c], xs[c], d, **SEG))
                l2, mx = oracle.err_metrics(y, oracle.fir_f64(taps[c], xs[c], d))
                assert l2 <= TOL and mx <= TOL, (c, l2, mx)
        with pytest.raises(fir.IfFirError, match="exceed"):
            mc.process_device([p.data_ptr() for p in dev_in], [p.data_ptr() for p in dev_in], n + 1)
        with pytest.raises(fir.IfFirError, match="rank 0 must pass"):
            mc.process_device(None, None, 16)
        mc.reset()                                                     # still usable after the errors
        out = torch.empty(2 * oracle.out_count(0, 64, d), dtype=torch.float32, device="cuda")
        outs = [out, out.clone(), out.clone()]
        assert mc.process_device([p.data_ptr() for p in dev_in], [o.data_ptr() for o in outs], 64) == 16


def test_multi_channel_front_chunked_equals_unchunked(fir, oracle, torch_cuda):
    """The multi-channel front moves and filters a call in chunks (multiples of the context's unit = lcm of ITS filter's
    overlap-save block advance and twice the decimation; a request is rounded to the nearest multiple) and the contexts keep a whole block overlap of
    history, so the blocks of a chunked call start where those of the unsplit call start and see the same samples: ANY
    chunking is bit-identical to no chunking, at ANY decimation phase (round 3: where the block grid follows the phase,
    decimation 4, the first chunk of an off-phase call is that many samples longer, so that no block of a chunk reaches
    past the chunk's end; the pieces then start at odd sample offsets, which the overlap-save kernel accepts).  One rank,
    three channels, two calls; D = 11 (round 3: chunks of 11 x 215040 samples; round 4: 5 x 42 240, the multiple of lcm(3840, 22)
    nearest to the request) on a longer stream."""
    torch = torch_cuda
    cases = [(255, 4, 600_000, 1_000_008), (1023, 1, 600_001, 1_000_008), (511, 3, 600_001, 1_000_008),
             (2047, 4, 600_003, 1_000_008), (255, 4, 600_002, 1_000_008), (255, 4, 600_001, 1_000_007),
             (255, 7, 600_003, 1_000_008), (3075, 4, 600_001, 1_000_008), (255, 8, 600_003, 1_000_008), (255, 2, 600_001, 1_000_008), (255, 32, 600_007, 1_000_008),
             (1023, 16, 600_005, 1_000_008), (255, 12, 600_005, 1_000_008), (255, 24, 600_007, 1_000_008), (1023, 40, 600_011, 1_000_008),
             (255, 6, 600_001, 1_000_008), (511, 10, 600_003, 1_000_008),
             (127, 1, 600_001, 1_000_008), (127, 3, 600_002, 1_000_008), (65, 5, 600_003, 1_000_007), (129, 1, 600_000, 1_000_008),
             (255, 11, 2_500_003, 7_400_000)]
    for t, d, first, n in cases:
        cuts = [0, first, n]
        nch = 3 if n < 2_000_000 else 1
        taps = np.stack([fir.bpf_design(t, 0.15, 0.25), fir.bpf_design(t, 0.02, 0.08), fir.bpf_design(t, 0.3, 0.45)][:nch])
        dev_in = [torch.from_numpy(oracle.synth_iq(n, 40 + c)).cuda() for c in range(nch)]
        results = {}
        with fir.IfFirMc(taps, d, n) as mc:
            adv = 2048 if t > 3073 else 3840 if t <= 257 else 3584 if t <= 513 else 3072 if t <= 1025 else 2048
            if d % 2 == 1 and d % 3 == 0 and t <= 767:      # the odd-decimation kernel: blocks of 3 x 1024 samples
                need = (t - 1 + 2 + 2) // 3
                adv = 3 * (1024 - 64 * (2 if need <= 128 else 4))
            for chunk in (fir.MC_NEVER_SPLIT, fir.MC_CHUNK_UNIT, 2 * fir.MC_CHUNK_UNIT, 100_000, 0):
                mc.set_chunk_samples(chunk)
                eff, unit = mc.get_chunk_samples()
                assert unit == np.lcm(adv, 2 * d), (t, d, unit)
                if chunk in (fir.MC_NEVER_SPLIT, 0):
                    assert eff == 0                  # one rank moves nothing: its calls are not split unless asked to
                else:
                    assert eff % unit == 0 and eff >= unit and (abs(eff - chunk) <= unit // 2 or eff == unit), (t, d, chunk, eff)
                mc.reset()
                parts = [[] for _ in range(nch)]
                for a, b in zip(cuts[:-1], cuts[1:]):
                    m_exp = oracle.out_count(a, b - a, d)
                    outs = [torch.full((2 * m_exp + 8,), 7.0, dtype=torch.float32, device="cuda") for _ in range(nch)]
                    pieces = [x[2 * a:2 * b].clone() for x in dev_in]
                    torch.cuda.synchronize()
                    assert mc.process_device([p.data_ptr() for p in pieces], [o.data_ptr() for o in outs], b - a) == m_exp
                    for c in range(nch):
                        o = outs[c].cpu().numpy()
                        assert np.all(o[2 * m_exp:] == 7.0)
                        parts[c].append(o[:2 * m_exp])
                results[chunk] = [np.concatenate(p) for p in parts]
        ref = results[fir.MC_NEVER_SPLIT]
        for chunk, res in results.items():
            for c in range(nch):
                assert np.array_equal(res[c], ref[c]), (t, d, first, chunk, c)
        l2, mx = oracle.err_metrics(ref[nch - 1], oracle.fir_f64(taps[nch - 1], dev_in[nch - 1].cpu().numpy(), d))
        assert l2 <= TOL and mx <= TOL, (t, d, l2, mx)


def test_multi_channel_front_splits_calls_on_the_overlap_save_backend_only(fir, oracle, torch_cuda):
    """the chunk table (block grid, phase shift of the first chunk, sample-aligned pieces) is the overlap-save
    backend's.  A channel switched to another backend is refused when the call would be split -- before anything runs --
    and works unsplit."""
    torch = torch_cuda
    t, d, n = 255, 4, 700_001
    taps = np.stack([fir.bpf_design(t, 0.15, 0.25), fir.bpf_design(t, 0.02, 0.08)])
    dev_in = [torch.from_numpy(oracle.synth_iq(n, 60 + c)).cuda() for c in range(2)]
    for fmt in (fir.INPUT_F32, fir.INPUT_I16):
        with fir.IfFirMc(taps, d, n) as mc:
            if fmt == fir.INPUT_I16:
                mc.set_input_format(fmt)
                src = [torch.from_numpy(np.random.default_rng(c).integers(-3000, 3000, 2 * n, dtype=np.int16)).cuda() for c in range(2)]
            else:
                src = dev_in
            mc.set_backend(fir.BACKEND_HIP_GENERIC)
            mc.set_chunk_samples(fir.MC_CHUNK_UNIT)
            m = oracle.out_count(0, n, d)
            outs = [torch.zeros(2 * m, dtype=torch.float32, device="cuda") for _ in range(2)]
            # an off-phase stream position first (one unsplit call of 3 samples), then a call that would be split
            mc.set_chunk_samples(fir.MC_NEVER_SPLIT)
            assert mc.process_device([x.data_ptr() for x in src], [o.data_ptr() for o in outs], 3) == 1
            mc.set_chunk_samples(fir.MC_CHUNK_UNIT)
            with pytest.raises(fir.IfFirError, match="overlap-save backend"):
                mc.process_device([x.data_ptr() for x in src], [o.data_ptr() for o in outs], n)
            mc.set_chunk_samples(fir.MC_NEVER_SPLIT)
            mc.reset()
            assert mc.process_device([x.data_ptr() for x in src], [o.data_ptr() for o in outs], n) == m
            xin = src[1].cpu().numpy()
            xin = xin.astype(np.float32) * np.float32(2.0 ** -15) if fmt == fir.INPUT_I16 else xin
            l2, mx = oracle.err_metrics(outs[1].cpu().numpy(), oracle.fir_f64(taps[1], xin, d))
            assert l2 <= TOL and mx <= TOL, (fmt, l2, mx)


def test_multi_channel_front_loopback_over_the_real_rccl(fir, oracle, torch_cuda):
    """The multi-channel front's whole transfer protocol over the REAL librccl on one GPU (round 3): with IF_FIR_MC_LOOPBACK=N the
    development library lets one process play all ranks of an N-rank world over a one-rank communicator -- every send of
    the plan is matched by its receive in the same group, peer = itself.  Chunks, grouped transfers, the two staging slots,
    events between the transfer and filter streams, the status word and the polling wait all run as between GPUs; only the
    wire is missing.  Results: bit-identical to the same calls without any transport, at decimation phases != 0 too."""
    torch = torch_cuda
    t, d, nch = 255, 4, 7        # (7 channels over 3 virtual ranks: ranks with 3, 2 and 2 channels -- the order of a group's
    cuts = [0, 600_001, 1_300_009]  #  sends and receives matters then: RCCL pairs them up in posting order)
    n = cuts[-1]
    taps = np.stack([fir.bpf_design(t, 0.02 + 0.05 * c, 0.08 + 0.05 * c) for c in range(nch)])
    dev_in = [torch.from_numpy(oracle.synth_iq(n, 60 + c)).cuda() for c in range(nch)]

    def run(loopback):
        if loopback:
            os.environ["IF_FIR_MC_LOOPBACK"] = str(loopback)     # the number of virtual ranks
        try:
            mc = fir.IfFirMc(taps, d, n, dev=True)
        except fir.IfFirError as e:
            if loopback and "librccl" in str(e):
                pytest.skip("librccl cannot be opened here: %s" % e)
            raise
        finally:
            os.environ.pop("IF_FIR_MC_LOOPBACK", None)
        with mc:
            mc.set_chunk_samples(fir.MC_CHUNK_UNIT)      # 215 040-sample chunks: 3 + 4 chunks in the two calls
            parts = [[] for _ in range(nch)]
            for a, b in zip(cuts[:-1], cuts[1:]):
                m_exp = oracle.out_count(a, b - a, d)
                outs = [torch.full((2 * m_exp + 8,), 7.0, dtype=torch.float32, device="cuda") for _ in range(nch)]
                pieces = [x[2 * a:2 * b].clone() for x in dev_in]
                torch.cuda.synchronize()
                assert mc.process_device([p.data_ptr() for p in pieces], [o.data_ptr() for o in outs], b - a) == m_exp
                for c in range(nch):
                    o = outs[c].cpu().numpy()
                    assert np.all(o[2 * m_exp:] == 7.0)
                    parts[c].append(o[:2 * m_exp])
            return [np.concatenate(p) for p in parts]
    plain = run(0)
    for vranks in (2, 3, 4):
        looped = run(vranks)
        for c in range(nch):
            assert np.array_equal(plain[c], looped[c]), (vranks, c)
    l2, mx = oracle.err_metrics(looped[1], oracle.fir_f64(taps[1], dev_in[1].cpu().numpy(), d))
    assert l2 <= TOL and mx <= TOL, (l2, mx)


def test_overlap_save_backend_takes_sample_aligned_pointers(fir, oracle, torch_cuda):
    """The overlap-save kernel moves one sample per lane: input and output pointers need only be aligned to a sample (8
    bytes; 4 for int16 input), which the chunked multi-channel front relies on; the results are those of aligned buffers."""
    torch = torch_cuda
    taps = fir.bpf_design(255)
    n = 300_001
    x = oracle.synth_iq(n + 1, 77)
    xd = torch.from_numpy(x).cuda()
    for d in (1, 4, 5):
        with fir.IfFir(taps, d, 0) as f:
            assert f.get_backend() == fir.BACKEND_HIP_FFT
            m = f.out_count(n)
            ref = torch.empty(2 * m, dtype=torch.float32, device="cuda")
            aligned = xd[2:2 * n + 2].clone()
            f.process_device(aligned.data_ptr(), ref.data_ptr(), n)
            f.reset()
            out = torch.full((2 * m + 6,), 9.0, dtype=torch.float32, device="cuda")
            assert f.process_device(xd.data_ptr() + 8, out.data_ptr() + 8, n) == m       # both 8 mod 16
            f.synchronize()
            assert torch.equal(out[2:2 * m + 2], ref) and bool((out[:2] == 9.0).all()) and bool((out[2 * m + 2:] == 9.0).all())
            with pytest.raises(fir.IfFirError, match="aligned"):
                f.process_device(xd.data_ptr() + 4, out.data_ptr(), n)
    xi = (np.clip(x, -1, 1) * 20000).astype(np.int16)
    xid = torch.from_numpy(xi).cuda()
    with fir.IfFir(taps, 4, 0) as f:
        f.set_input_format(fir.INPUT_I16)
        m = f.out_count(n)
        ref = torch.empty(2 * m, dtype=torch.float32, device="cuda")
        aligned = xid[2:2 * n + 2].clone()
        f.process_device(aligned.data_ptr(), ref.data_ptr(), n)
        f.reset()
        out = torch.empty(2 * m, dtype=torch.float32, device="cuda")
        f.process_device(xid.data_ptr() + 4, out.data_ptr(), n)                           # 4 mod 16
        f.synchronize()
        assert torch.equal(out, ref)


def test_multi_channel_bootstrap_id(fir, gpu_ok):
    """if_fir_mc_unique_id(): librccl is opened on demand and hands out a bootstrap id (the >1-rank transfers need
    more than the one GPU of this box; their rank/peer bookkeeping is the same c mod world rule the gloo tests cover)."""
    a, b = fir.mc_unique_id(), fir.mc_unique_id()
    assert len(a) == fir.MC_ID_BYTES and any(a) and a != b


@pytest.mark.parametrize("t,d,backend", [(255, 4, "fft"), (255, 1, "fft"), (1023, 4, "fft"), (127, 1, "fft"),
                                         (31, 3, "generic"), (255, 4, "generic")])
def test_nco_fused_into_the_filter(fir, oracle, t, d, backend):
    """SPEC §3.2: NCO mix ahead of the filter = complex taps + output rotation inside the kernels.
    Against the float64 oracle that mixes every input sample by the definition; one call and ragged pieces (the phase
    follows the absolute sample index, the history holds unmixed samples)."""
    rng = np.random.default_rng(1000 + t + d)
    taps = fir.bpf_design(t, 0.0, 0.06) if t >= 3 else np.array([1.0], np.float32)     # low-pass prototype
    n = 60_011
    x = np.concatenate([oracle.synth_iq(n // 2, 17), rng.standard_normal(2 * (n - n // 2)).astype(np.float32)])
    f = 0.2003
    pw = oracle.nco_phase_word(f)
    ref = oracle.fir_nco_f64(taps, x, d, pw)
    with fir.IfFir(taps, d, n) as flt:
        flt.set_backend(fir.BACKEND_HIP_FFT if backend == "fft" else fir.BACKEND_HIP_GENERIC)
        flt.set_nco(f)
        assert abs(flt.get_nco() - pw / 2.0 ** 32) < 1e-15 and abs(flt.get_nco() - f) <= 2.0 ** -33
        y = flt.process(x)
        l2, mx = oracle.err_metrics(y, ref)
        assert l2 <= TOL and mx <= TOL, (l2, mx)
        flt.reset()
        cuts = [0, 1, 2, 7, 1021, 3841, 3840 * 2 + 6, 20_001, 40_003, n]
        parts = [flt.process(x[2 * a:2 * b]) for a, b in zip(cuts[:-1], cuts[1:])]
        l2, mx = oracle.err_metrics(np.concatenate(parts), ref)
        assert l2 <= TOL and mx <= TOL, (l2, mx)
        # negative frequency, then off again: the plain path comes back bit for bit
        flt.reset()
        flt.set_nco(-0.31)
        l2, mx = oracle.err_metrics(flt.process(x), oracle.fir_nco_f64(taps, x, d, oracle.nco_phase_word(-0.31)))
        assert l2 <= TOL and mx <= TOL, (l2, mx)
        flt.reset()
        flt.set_nco(0.0)
        y0 = flt.process(x)
    with fir.IfFir(taps, d, n) as flt:
        flt.set_backend(fir.BACKEND_HIP_FFT if backend == "fft" else fir.BACKEND_HIP_GENERIC)
        assert np.array_equal(flt.process(x), y0)


def test_nco_golden_int16_complex_taps_and_refusals(fir, oracle):
    gold = np.load(GOLD)
    pw = int(gold["nco_phase_word"][0])
    h, x = gold["taps_lp_255"], gold["x"]
    for d in (1, 4):
        with fir.IfFir(h, d, 4096) as f:
            f.set_nco(0.2)
            assert f.get_backend() == fir.BACKEND_HIP_FFT and round(f.get_nco() * 2 ** 32) == pw
            l2, mx = oracle.err_metrics(f.process(x), gold["ynco_T255_D%d" % d])     # numpy/scipy-made fixture
            assert l2 <= TOL and mx <= TOL, (d, l2, mx)
    # int16 input + NCO (both halves of f-1 together), overlap-save and generic
    n = 50_003
    xi = np.clip(np.round(oracle.synth_iq(n, 19) * 16384.0), -32768, 32767).astype(np.int16)
    xf = xi.astype(np.float32) * np.float32(2.0 ** -15)
    ref = oracle.fir_nco_f64(h, xf, 4, oracle.nco_phase_word(-0.123))
    with fir.IfFir(h, 4, n) as f:
        f.set_input_format(fir.INPUT_I16)
        f.set_nco(-0.123)
        for b in (fir.BACKEND_HIP_FFT, fir.BACKEND_HIP_GENERIC):
            f.set_backend(b)
            f.reset()
            parts = [f.process(xi[2 * a:2 * b2]) for a, b2 in ((0, 20_001), (20_001, n))]
            l2, mx = oracle.err_metrics(np.concatenate(parts), ref)
            assert l2 <= TOL and mx <= TOL, (b, l2, mx)
    # complex taps + NCO: the taps are rotated as well
    g = fir.bpf_design_complex(255, 0.1, 0.08)
    ref = oracle.fir_nco_f64(g, x, 4, oracle.nco_phase_word(0.05), complex_taps=True)
    with fir.IfFir(g, 4, 4096, complex_taps=True) as f:
        f.set_nco(0.05)
        l2, mx = oracle.err_metrics(f.process(x), ref)
        assert l2 <= TOL and mx <= TOL, (l2, mx)
    # kernels with real-tap arithmetic refuse; the context keeps working without NCO
    with fir.IfFir(fir.bpf_design(255), 4, 4096, backend=fir.BACKEND_HIP_DIRECT) as f:
        with pytest.raises(fir.IfFirError, match="real-tap"):
            f.set_nco(0.1)
        assert f.get_nco() == 0.0
        assert np.array_equal(f.process(x), oracle.fir_f32fma(fir.bpf_design(255), x, 4, **SEG))
        with pytest.raises(fir.IfFirError):
            f.set_nco(0.75)                       # out of range
    with fir.IfFir(h, 4, 4096) as f:
        f.set_nco(0.1)
        with pytest.raises(fir.IfFirError):
            f.set_backend(fir.BACKEND_HIP_DIRECT)  # and the other way round


@pytest.mark.parametrize("t", [255, 1023, 63, 511])
def test_uniform_filter_bank(fir, oracle, torch_cuda, t):
    """if_fir_channelizer_process_device = the channels' NCO + prototype + decimate-by-4 results from one
    pass over the input.  Every channel against the float64 oracle (NCO phase word slot * 2^28), in ragged pieces (odd
    cuts: every decimation phase and every sample-index residue mod 16 at a call boundary), on the normal grid and
    through the run queue of a one-workgroup launch."""
    torch = torch_cuda
    taps = fir.bpf_design(t, 0.0, 0.03)
    n = 300_007 if t == 255 else 90_011
    x = oracle.synth_iq(n, 41)
    xd = torch.from_numpy(x).cuda()
    slots = [0, 3, 5, 15, 8, 3, 10]
    refs = [oracle.fir_nco_f64(taps, x, 4, (s << 28) & 0xFFFFFFFF) for s in slots]
    cuts = [0, 1, 6, 4103, 40_001, 40_004, n]
    with fir.IfFir(taps, 4, 0, dev=True) as f:
        f.set_backend(fir.BACKEND_HIP_FFT)
        for tuning in (0, 2001):
            f.set_tuning(tuning)
            f.reset()
            parts = [[] for _ in slots]
            for a, b in zip(cuts[:-1], cuts[1:]):
                m_exp = oracle.out_count(a, b - a, 4)
                outs = [torch.full((2 * m_exp + 8,), 3.0, dtype=torch.float32, device="cuda") for _ in slots]
                piece = xd[2 * a:2 * b].clone()
                torch.cuda.synchronize()
                assert f.channelizer_process_device(slots, piece.data_ptr(), [o.data_ptr() for o in outs], b - a) == m_exp
                f.synchronize()
                for c in range(len(slots)):
                    o = outs[c].cpu().numpy()
                    assert np.all(o[2 * m_exp:] == 3.0)
                    parts[c].append(o[:2 * m_exp])
            for c, s in enumerate(slots):
                y = np.concatenate(parts[c])
                l2, mx = oracle.err_metrics(y, refs[c])
                assert l2 <= TOL and mx <= TOL, (t, tuning, s, l2, mx)
            assert np.array_equal(np.concatenate(parts[1]), np.concatenate(parts[5]))      # the same slot twice
        # refusals: wrong context kinds, bad slots
        with pytest.raises(fir.IfFirError, match="slot"):
            f.channelizer_process_device([16], xd.data_ptr(), [xd.data_ptr()], 16)
        f.set_nco(0.1)
        with pytest.raises(fir.IfFirError, match="no NCO"):
            f.channelizer_process_device([1], xd.data_ptr(), [xd.data_ptr()], 16)   # (decimation 4 only: 8 / 16 take one)
    with fir.IfFir(taps, 1, 0) as f:
        with pytest.raises(fir.IfFirError, match="multiple of 4"):
            f.channelizer_process_device([1], xd.data_ptr(), [xd.data_ptr()], 16)


@pytest.mark.parametrize("d,t,i16", [(16, 255, False), (16, 1023, False), (16, 127, False), (16, 511, True), (16, 2047, False),
                                      (8, 255, False), (8, 1023, False), (8, 63, False), (8, 511, True), (8, 2047, False),
                                      (8, 3073, False), (16, 3073, True)])   # (3073 taps: the longest prototype, 48 overlap rows)
def test_filter_bank_at_decimation_8_and_16(fir, oracle, torch_cuda, monkeypatch, d, t, i16):
    """the bank at decimation 16 = the rate of an fs/16-wide channel (the kernel computes ALL
    16 slots from one forward transform: the 16-way alias fold of slot s is output s of one 16-point transform per group)
    and at decimation 8 = 2x oversampled channels (per channel, two channels per 512-point inverse), DESIGN §3.7.  Every
    channel against the float64 NCO oracle (phase word slot * 2^28), ragged pieces (every phase at a call boundary),
    float32 and int16 input, the run queue of a one-workgroup launch, all 16 slots and subsets (decimation 8: repeats and odd
    counts too), nothing written outside the wanted buffers."""
    torch = torch_cuda
    monkeypatch.setenv("IF_FIR_DEBUG", "1")   # (the development launch 4096 below)
    taps = fir.bpf_design(t, 0.0, 0.02 if d == 16 else 0.04)
    n = 400_011 if t <= 255 else 150_013
    if i16:
        xi = np.clip(np.round(oracle.synth_iq(n, 47) * 12000.0), -32768, 32767).astype(np.int16)
        x = xi.astype(np.float32) * np.float32(2.0 ** -15)
        xd = torch.from_numpy(xi).cuda()
        cuts = [0, 4, 50_004, 120_008, n]         # int16 pieces start on 4-sample boundaries
    else:
        x = oracle.synth_iq(n, 47)
        xd = torch.from_numpy(x).cuda()
        cuts = [0, 1, 6, 4103, 40_001, 40_018, 120_007, n]
    refs = {s: oracle.fir_nco_f64(taps, x, d, (s << 28) & 0xFFFFFFFF) for s in range(16)}
    subsets = ((list(range(16)), 0), ([5, 0, 15, 8, 3, 10, 1, 14], 0), ([7, 2], 2001))
    if d == 8:
        # round 4: a parity (even / odd slots) with >= 4 channels, no slot twice, runs the ALL-SLOTS form (two 8-point transforms
        # per group give the eight slots of that parity; both parities: ONE launch over virtual blocks 2 b + parity; the rest keep the
        # per-channel form: up to two launches a call)
        subsets = ((list(range(16)), 0), ([5, 0, 15, 8, 3, 10, 3], 0), ([7], 2001),
                   ([1, 5, 9, 13, 2], 0),                       # odd slots all-slots + one even channel per-channel
                   ([0, 2, 4, 6, 8, 10, 12, 14], 2001),         # even slots, one-workgroup grid (run queue)
                   ([3, 1, 15, 13, 11, 9, 7, 5, 0, 4], 0),      # all odd slots + two even ones
                   ([14, 3, 8, 6, 0], 1004096),                 # (development launch 4096: per-channel form although 4 even)
                   (list(range(16)), 1008192),                  # (development launch 8192: the two parities as two launches, not one)
                   ([1, 0, 3, 2, 5, 4, 7, 6, 15], 2001))        # both parities in one launch (virtual blocks), one-workgroup grid
    with fir.IfFir(taps, d, n, dev=True) as f:
        if i16:
            f.set_input_format(fir.INPUT_I16)
        assert f.get_backend() == fir.BACKEND_HIP_FFT
        for slots, tuning in subsets:
            f.set_tuning(tuning)
            f.reset()
            parts = [[] for _ in slots]
            for a, b in zip(cuts[:-1], cuts[1:]):
                m_exp = oracle.out_count(a, b - a, d)
                outs = [torch.full((2 * m_exp + 8,), 3.0, dtype=torch.float32, device="cuda") for _ in slots]
                piece = xd[2 * a:2 * b].clone()
                torch.cuda.synchronize()
                assert f.channelizer_process_device(slots, piece.data_ptr(), [o.data_ptr() for o in outs], b - a) == m_exp
                f.synchronize()
                for c in range(len(slots)):
                    o = outs[c].cpu().numpy()
                    assert np.all(o[2 * m_exp:] == 3.0)
                    parts[c].append(o[:2 * m_exp])
            for c, sl in enumerate(slots):
                l2, mx = oracle.err_metrics(np.concatenate(parts[c]), refs[sl])
                assert l2 <= TOL and mx <= TOL, (d, t, i16, tuning, sl, l2, mx)
        assert f.debug_queue_faults() == 0
        # the context still filters one channel the ordinary way (selecting store), same stream semantics
        f.set_tuning(0)
        f.reset()
        y = f.process(xi if i16 else x)
        l2, mx = oracle.err_metrics(y, oracle.fir_f64(taps, x, d))
        assert l2 <= TOL and mx <= TOL, (l2, mx)
        if d == 16:
            with pytest.raises(fir.IfFirError, match="twice"):
                f.channelizer_process_device([3, 3], xd.data_ptr(), [xd.data_ptr(), xd.data_ptr()], 16)


@pytest.mark.parametrize("d,i16", [(16, False), (8, False), (16, True), (8, True)])
def test_filter_bank_with_a_common_fine_offset(fir, oracle, torch_cuda, d, i16):
    """The context's NCO shifts the whole fs/16 slot grid (decimation 8 and 16): channel c is centred at slot/16 + f_nco.
    Every channel against the float64 NCO oracle at that frequency, ragged pieces."""
    torch = torch_cuda
    taps = fir.bpf_design(255, 0.0, 0.02)
    n = 200_007
    f0 = 0.0123
    if i16:
        xi = np.clip(np.round(oracle.synth_iq(n, 53) * 12000.0), -32768, 32767).astype(np.int16)
        x = xi.astype(np.float32) * np.float32(2.0 ** -15)
        xd = torch.from_numpy(xi).cuda()
        cuts = [0, 4, 50_004, n]
    else:
        x = oracle.synth_iq(n, 53)
        xd = torch.from_numpy(x).cuda()
        cuts = [0, 1, 40_001, 40_018, n]
    # (decimation 8, round 4: a slot parity with >= 4 channels takes the all-slots launch with the NCO as well -- both parities: one launch --,
    # the rest the per-channel general form)
    for slots in (([0, 5, 9, 15, 2], [1, 3, 5, 7, 9, 2, 4, 6, 8], [13, 5, 9, 1, 0]) if d == 8 else ([0, 5, 9, 15, 2, 12],)):
        with fir.IfFir(taps, d, n) as f:
            if i16:
                f.set_input_format(fir.INPUT_I16)
            f.set_nco(f0)
            word = oracle.nco_phase_word(f.get_nco())
            parts = [[] for _ in slots]
            for a, b in zip(cuts[:-1], cuts[1:]):
                m_exp = oracle.out_count(a, b - a, d)
                outs = [torch.full((2 * m_exp + 8,), 3.0, dtype=torch.float32, device="cuda") for _ in slots]
                piece = xd[2 * a:2 * b].clone()
                torch.cuda.synchronize()
                assert f.channelizer_process_device(slots, piece.data_ptr(), [o.data_ptr() for o in outs], b - a) == m_exp
                f.synchronize()
                for c in range(len(slots)):
                    o = outs[c].cpu().numpy()
                    assert np.all(o[2 * m_exp:] == 3.0)
                    parts[c].append(o[:2 * m_exp])
            for c, sl in enumerate(slots):
                ref = oracle.fir_nco_f64(taps, x, d, ((sl << 28) + word) & 0xFFFFFFFF)
                l2, mx = oracle.err_metrics(np.concatenate(parts[c]), ref)
                assert l2 <= TOL and mx <= TOL, (d, i16, slots, sl, l2, mx)
    with fir.IfFir(taps, 4, n) as f:
        f.set_nco(f0)
        with pytest.raises(fir.IfFirError, match="multiple of 4"):
            f.channelizer_process_device([1], xd.data_ptr(), [xd.data_ptr()], 16)


@pytest.mark.parametrize("t,i16,d", [(255, False, 8), (1023, False, 8), (127, True, 8), (511, False, 8),
                                     (255, False, 16), (1023, True, 16), (63, False, 16), (2047, False, 16),
                                     (255, False, 4), (127, True, 4), (1023, False, 4), (3073, False, 4),
                                     # every other multiple of 4: the tail of the largest of 16 / 8 / 4 dividing it, keeping every (D / F)-th output
                                     (255, False, 32), (511, True, 64), (255, False, 12), (1023, False, 24), (255, True, 48), (127, False, 20)])
def test_filter_bank_channels_at_arbitrary_centre_frequencies(fir, oracle, torch_cuda, t, i16, d):
    """Round 4: channels at ARBITRARY centres from one pass (decimation 8, and 16 = the channel rate, four channels per
    small inverse; and 4, one 1024-point inverse per channel): the prototype moved up by the multiple
    of fs/4096 nearest to the wanted centre (a shift of the overlap-save transform's bins: per lane another table row, the
    lanes rotated), mixed down by the exact centre.  Centres ON the grid: every channel against the float64 NCO oracle at that
    frequency = what C contexts with if_fir_set_nco(f_c) compute (also checked against one such context on the GPU).  Centres
    OFF the grid: against the by-definition float64 form  y = exp(-j 2 pi f a) (h exp(+j 2 pi g k) * x),  g = round(4096 f) / 4096.
    Ragged pieces (history, phase and sample index carried), odd channel counts, negative centres, float32 / int16."""
    torch = torch_cuda
    taps = fir.bpf_design(t, 0.0, 0.02)
    n = 120_011
    if i16:
        xi = np.clip(np.round(oracle.synth_iq(n, 57) * 12000.0), -32768, 32767).astype(np.int16)
        x = xi.astype(np.float32) * np.float32(2.0 ** -15)
        xd = torch.from_numpy(xi).cuda()
        cuts = [0, 8, 50_008, n]
    else:
        x = oracle.synth_iq(n, 57)
        xd = torch.from_numpy(x).cuda()
        cuts = [0, 1, 40_001, 40_018, n]
    on_grid = [b / 4096.0 for b in (0, 1, 255, 256, 257, 300, 819, 2047, -2048, -1, -333, 1638, 77)]       # 13 channels
    off_grid = [0.2, -0.123456789, 0.05 + 1.0 / 8192 - 1e-9, 0.3333333, 1e-7]
    for centres in (on_grid, off_grid):
        with fir.IfFir(taps, d, n) as f:
            if i16:
                f.set_input_format(fir.INPUT_I16)
            parts = [[] for _ in centres]
            for a, b in zip(cuts[:-1], cuts[1:]):
                m_exp = oracle.out_count(a, b - a, d)
                outs = [torch.full((2 * m_exp + 8,), 3.0, dtype=torch.float32, device="cuda") for _ in centres]
                piece = xd[2 * a:2 * b].clone()
                torch.cuda.synchronize()
                assert f.channelizer_process_device_freq(centres, piece.data_ptr(), [o.data_ptr() for o in outs], b - a) == m_exp
                f.synchronize()
                for c in range(len(centres)):
                    o = outs[c].cpu().numpy()
                    assert np.all(o[2 * m_exp:] == 3.0)
                    parts[c].append(o[:2 * m_exp])
        xc = x[0::2].astype(np.float64) + 1j * x[1::2].astype(np.float64)
        for c, fc in enumerate(centres):
            got = np.concatenate(parts[c])
            if centres is on_grid:
                ref = oracle.fir_nco_f64(taps, x, d, oracle.nco_phase_word(fc))
            else:
                g = np.round(fc * 4096.0) / 4096.0
                k = np.arange(t)
                hc = taps.astype(np.float64) * np.exp(2j * np.pi * g * k)
                y = np.convolve(xc, hc)[:n][::d]
                word = oracle.nco_phase_word(fc)
                aidx = np.arange(0, n, d, dtype=np.uint64)
                ph = ((np.uint64(word) * aidx) & np.uint64(0xFFFFFFFF)).astype(np.float64) / 2.0 ** 32
                y = y * np.exp(-2j * np.pi * ph)
                ref = np.empty(2 * y.size, dtype=np.float64)
                ref[0::2], ref[1::2] = y.real, y.imag
            l2, mx = oracle.err_metrics(got, ref)
            assert l2 <= TOL and mx <= TOL, (t, i16, fc, l2, mx)
    # one on-grid channel against a single-channel context with the NCO set to that centre (the C contexts of the claim)
    with fir.IfFir(taps, d, n) as f1:
        if i16:
            f1.set_input_format(fir.INPUT_I16)
        f1.set_nco(300 / 4096.0)
        y1 = f1.process(xi if i16 else x)
    with fir.IfFir(taps, d, n) as f:
        if i16:
            f.set_input_format(fir.INPUT_I16)
        out = torch.zeros(2 * oracle.out_count(0, n, d), dtype=torch.float32, device="cuda")
        f.channelizer_process_device_freq([300 / 4096.0], xd.data_ptr(), [out.data_ptr()], n)
        f.synchronize()
        l2, mx = oracle.err_metrics(out.cpu().numpy(), y1.astype(np.float64))
        assert l2 <= 2 * TOL and mx <= 2 * TOL, (l2, mx)
        # refusals: an NCO on the context, another decimation, a centre outside +-0.5
        with pytest.raises(fir.IfFirError, match="0.5"):
            f.channelizer_process_device_freq([0.6], xd.data_ptr(), [out.data_ptr()], 16)
        f.set_nco(0.01)
        with pytest.raises(fir.IfFirError, match="no NCO"):
            f.channelizer_process_device_freq([0.1], xd.data_ptr(), [out.data_ptr()], 16)
    with fir.IfFir(taps, 2, n) as f:
        with pytest.raises(fir.IfFirError, match="multiple of 4"):
            f.channelizer_process_device_freq([0.1], xd.data_ptr(), [xd.data_ptr()], 16)


def test_random_configurations_against_the_oracle(fir, oracle):
    """Sweep of random (taps, decimation, length, backend, piece cuts, input format, NCO) combinations: every result
    within SPEC tolerance of the float64 oracle, the bit-exact kernels equal to their order models."""
    rng = np.random.default_rng(int(os.environ.get("IF_FIR_TEST_SEED", "20261003")))   # other seeds: soak runs
    names = {fir.BACKEND_HIP_DIRECT: "direct", fir.BACKEND_HIP_TAPSPLIT: "tapsplit", fir.BACKEND_HIP_GENERIC: "generic",
             fir.BACKEND_HIP_FFT: "fft"}
    done = {k: 0 for k in names.values()}
    for case in range(60):
        t = int(rng.choice([1, 2, 3, 5, 16, 31, 32, 33, 63, 64, 65, 127, 128, 255, 256, 257, 258, 511, 777, 1023, 1025,
                            1026, 2047, 4096]))
        # (soak seeds also draw the decimations that run behind a tail keeping every sub-th output, round 3)
        d = int(rng.choice([1, 1, 2, 3, 4, 4, 5, 8, 16, 64] if "IF_FIR_TEST_SEED" not in os.environ else
                           [1, 1, 2, 3, 3, 4, 4, 5, 6, 8, 9, 10, 12, 15, 16, 20, 24, 27, 28, 48, 62, 63, 64]))   # (round 4: 9, 15, 27, 63)
        n = int(rng.integers(1, 30_000))
        taps = (rng.standard_normal(t) / np.sqrt(t)).astype(np.float32)
        x = rng.standard_normal(2 * n).astype(np.float32)
        nco = float(rng.uniform(-0.5, 0.5)) if rng.random() < 0.3 else 0.0
        i16 = rng.random() < 0.25
        if case % 6 == 0:      # the unrolled direct form exists for four (taps, decimation) pairs only: aim at them
            t, d, nco, i16 = int(rng.choice([127, 255])), int(rng.choice([1, 4])), 0.0, False
            taps = (rng.standard_normal(t) / np.sqrt(t)).astype(np.float32)
        if i16:
            xi = np.clip(np.round(x * 8000.0), -32768, 32767).astype(np.int16)
            x = xi.astype(np.float32) * np.float32(2.0 ** -15)
        choices = [fir.BACKEND_HIP_GENERIC]
        if not nco and not i16:
            choices.append(fir.BACKEND_HIP_TAPSPLIT)
            if t in (127, 255) and d in (1, 4):
                choices.append(fir.BACKEND_HIP_DIRECT)
        choices += [fir.BACKEND_HIP_FFT] * 2
        b = fir.BACKEND_HIP_DIRECT if case % 6 == 0 else int(rng.choice(choices))
        cuts = sorted(set([0, n] + [int(c) for c in rng.integers(0, n + 1, size=int(rng.integers(0, 4)))]))
        ref = oracle.fir_nco_f64(taps, x, d, oracle.nco_phase_word(nco)) if nco else oracle.fir_f64(taps, x, d)
        with fir.IfFir(taps, d, n) as f:
            if i16:
                f.set_input_format(fir.INPUT_I16)
            if nco:
                f.set_nco(nco)
            f.set_backend(b)
            src = xi if i16 else x
            y = np.concatenate([f.process(src[2 * a:2 * c]) for a, c in zip(cuts[:-1], cuts[1:])] or [np.zeros(0, np.float32)])
        assert y.shape == ref.shape, (case, t, d, n, names[b])
        if ref.size and np.any(ref):
            l2, mx = oracle.err_metrics(y, ref)
            assert l2 <= TOL and mx <= TOL, (case, t, d, n, names[b], nco, i16, cuts, l2, mx)
        if b in (fir.BACKEND_HIP_DIRECT, fir.BACKEND_HIP_GENERIC) and not nco:
            assert np.array_equal(y, oracle.fir_f32fma(taps, x, d, **SEG)), (case, t, d, n, names[b])
        done[names[b]] += 1
    if "IF_FIR_TEST_SEED" not in os.environ:   # (soak seeds draw the backends as they come; the default seed covers all four)
        assert all(v >= 3 for v in done.values()), done


def test_random_filter_bank_configurations_against_the_oracle(fir, oracle, torch_cuda):
    """Sweep of random filter-bank calls: decimation 4 / 8 / 16, random prototypes, random slot subsets (decimation 8:
    repeats too -- the routing between the all-slots launches and the per-channel form follows the subset), real and complex prototypes, channels at random
    centres on the fs/4096 grid (decimation 4 / 8 / 16 and other multiples of 4), a common fine offset (the context's NCO, decimation 8 / 16), float32 / int16, random
    piece cuts.  Every channel within SPEC tolerance of the float64 NCO oracle; nothing written past a channel's outputs."""
    torch = torch_cuda
    rng = np.random.default_rng(int(os.environ.get("IF_FIR_TEST_SEED", "20261004")) + 17)   # other seeds: soak runs
    schedule = [("slots", 4), ("slots", 8), ("slots", 16), ("freq", 4), ("freq", 8), ("freq", 16), ("nco", 8), ("nco", 16), ("allslots", 8),
                ("freq", 12), ("freq", 24), ("freq", 32), ("freq", 64), ("freq", 40),
                ("slots", 32), ("slots", 12), ("nco", 24), ("nco", 64)]   # every form in turn, the rest random
    for case in range(54):
        kind, d = schedule[case % len(schedule)]
        t = int(rng.choice([1, 2, 17, 63, 64, 65, 127, 255, 256, 257, 511, 777, 1023, 1025, 2047, 3073]))
        n = int(rng.integers(1, 40_000))
        ctaps = rng.random() < 0.3      # complex prototype taps (interleaved re, im)
        taps = (rng.standard_normal(2 * t if ctaps else t) / np.sqrt(t)).astype(np.float32)
        i16 = rng.random() < 0.3
        x = rng.standard_normal(2 * n).astype(np.float32)
        if i16:
            xi = np.clip(np.round(x * 8000.0), -32768, 32767).astype(np.int16)
            x = xi.astype(np.float32) * np.float32(2.0 ** -15)
        freq = kind.startswith("freq")
        nco = float(rng.uniform(-0.5, 0.5)) if kind.startswith("nco") else 0.0
        nch = int(rng.integers(1, 17))
        if kind == "allslots":   # at least four channels of one slot parity, no slot twice: that parity runs the all-slots launch
            par = int(rng.integers(0, 2))
            own = [int(v) for v in 2 * rng.permutation(8)[:int(rng.integers(4, 9))] + par]
            other = [int(v) for v in 2 * rng.permutation(8)[:int(rng.integers(0, 9))] + (1 - par)]
            slots = [int(v) for v in rng.permutation(own + other)]
            nch = len(slots)
        elif d == 16 or (d == 8 and rng.random() < 0.5):
            slots = [int(v) for v in rng.permutation(16)[:nch]]                 # each slot once
        else:
            slots = [int(v) for v in rng.integers(0, 16, size=nch)]              # repeats allowed (decimation 4 / 8)
        centres = [int(v) / 4096.0 for v in rng.integers(-2048, 2048, size=nch)]
        cuts = sorted(set([0, n] + [int(c) & (~3 if i16 else ~0) for c in rng.integers(0, n + 1, size=int(rng.integers(0, 4)))]))
        xd = torch.from_numpy(xi if i16 else x).cuda()
        with fir.IfFir(taps, d, n, complex_taps=ctaps) as f:
            if i16:
                f.set_input_format(fir.INPUT_I16)
            if nco:
                f.set_nco(nco)
            word = oracle.nco_phase_word(f.get_nco()) if nco else 0
            parts = [[] for _ in range(nch)]
            for a, b in zip(cuts[:-1], cuts[1:]):
                m_exp = oracle.out_count(a, b - a, d)
                outs = [torch.full((2 * m_exp + 8,), 3.0, dtype=torch.float32, device="cuda") for _ in range(nch)]
                piece = xd[2 * a:2 * b].clone()
                torch.cuda.synchronize()
                ptrs = [o.data_ptr() for o in outs]
                if freq:
                    assert f.channelizer_process_device_freq(centres, piece.data_ptr(), ptrs, b - a) == m_exp
                else:
                    assert f.channelizer_process_device(slots, piece.data_ptr(), ptrs, b - a) == m_exp
                f.synchronize()
                for c in range(nch):
                    o = outs[c].cpu().numpy()
                    assert np.all(o[2 * m_exp:] == 3.0), (case, d, t, n, c)
                    parts[c].append(o[:2 * m_exp])
        for c in range(nch):
            pw = oracle.nco_phase_word(centres[c]) if freq else ((slots[c] << 28) + word) & 0xFFFFFFFF
            ref = oracle.fir_nco_f64(taps, x, d, pw, complex_taps=ctaps)
            got = np.concatenate(parts[c]) if parts[c] else np.zeros(0, np.float32)
            assert got.shape == ref.shape
            if ref.size and np.any(ref):
                l2, mx = oracle.err_metrics(got, ref)
                assert l2 <= TOL and mx <= TOL, (case, kind, d, t, n, i16, ctaps, nco, slots, centres[c], cuts, l2, mx)


def test_contexts_on_concurrent_threads(fir, oracle):
    """include/if_fir.h: distinct contexts may be used from distinct threads.  Four threads, each with its own context
    and backend, filter their own stream in pieces at the same time (ctypes releases the GIL during the calls)."""
    import threading
    n = 400_003
    jobs = [(255, 4, fir.BACKEND_HIP_FFT), (255, 4, fir.BACKEND_HIP_DIRECT), (1023, 1, fir.BACKEND_HIP_FFT),
            (63, 3, fir.BACKEND_HIP_TAPSPLIT)]
    inputs = [oracle.synth_iq(n, 50 + k) for k in range(len(jobs))]
    results, errors = [None] * len(jobs), []

    def work(k):
        try:
            t, d, b = jobs[k]
            with fir.IfFir(fir.bpf_design(t), d, n, backend=b) as f:
                for _ in range(3):                                   # repeat: more time for the threads to overlap
                    f.reset()
                    cuts = [0, 100_001, 250_000, n]
                    results[k] = np.concatenate([f.process(inputs[k][2 * a:2 * c]) for a, c in zip(cuts[:-1], cuts[1:])])
        except Exception as e:   # noqa: BLE001
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(len(jobs))]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for k, (t, d, b) in enumerate(jobs):
        l2, mx = oracle.err_metrics(results[k], oracle.fir_f64(fir.bpf_design(t), inputs[k], d))
        assert l2 <= TOL and mx <= TOL, (k, l2, mx)


@pytest.mark.parametrize("t,d", [(255, 2), (255, 3), (1023, 8), (63, 5), (257, 16), (1025, 64), (255, 7), (255, 8), (2047, 16),
                                 (3073, 8), (1023, 2), (2047, 2), (3073, 2), (31, 2), (255, 32), (1023, 32), (127, 64), (3073, 64),
                                 (255, 12), (1023, 20), (255, 24), (513, 28), (2047, 40), (255, 48), (3073, 56), (255, 60), (127, 44),
                                 (3073, 12), (255, 6), (3075, 12), (1023, 10), (255, 62), (513, 14), (3073, 30), (3075, 6),
                                 (4095, 2), (4095, 16), (3333, 4), (4001, 20),
                                 # round 4: decimation 3, 9, 15, ... on the odd-decimation kernel (blocks of 3 x 1024 samples; 2, 4 or 8
                                 # dropped output rows: <= 383 / 767 taps), longer filters and decimation 5 / 25 on the selecting store
                                 (255, 9), (383, 3), (385, 3), (767, 15), (769, 3), (1023, 3), (1535, 3), (1537, 3), (255, 21), (255, 63),
                                 (3, 3), (5, 9), (127, 27), (511, 33), (255, 25), (129, 45)])
def test_fft_backend_any_decimation(fir, oracle, t, d):
    """Decimations other than 1 and 4 on the overlap-save backend: the full-rate kernel keeps every D-th output (one
    64-bit division per block and lane, an exact multiply-shift per row); decimation 2 (frequency-domain fold + 2048-point
    inverse) and 8 / 16 / 32 / 64 (the one-channel filter-bank route; 32 and 64 keep every 2nd / 4th output of the
    decimate-by-16 tail) have their own tails since round 3 (development variant 3000 = the selecting store for them too); so
    have the multiples of 4 and 8 (12, 20, ..., 60 behind the decimate-by-4 tail, 24, 40, 48, 56 behind the decimate-by-8 one: the
    tail keeps every 3rd, 5th, ... output), and 6, 10, ..., 62 behind the decimate-by-2 tail: every even decimation has a tail; round 4:
    decimation 3, 9, 15, ..., 63 on the odd-decimation kernel (three forward 1024-point transforms of the phase streams, one inverse;
    variant 3000 = the selecting store with the ordinary tables, rebuilt when the variant changes).  One
    call, ragged pieces (every decimation phase at a call boundary), the run queue on a one-workgroup grid, NCO (always
    the selecting store) and int16 input on top."""
    rng = np.random.default_rng(7000 + t + d)
    taps = fir.bpf_design(t)
    n = 120_011
    x = np.concatenate([oracle.synth_iq(n // 2, 23), rng.standard_normal(2 * (n - n // 2)).astype(np.float32)])
    ref = oracle.fir_f64(taps, x, d)
    with fir.IfFir(taps, d, n, dev=True) as f:
        f.set_backend(fir.BACKEND_HIP_FFT)
        y = f.process(x)
        assert y.shape == ref.shape
        l2, mx = oracle.err_metrics(y, ref)
        assert l2 <= TOL and mx <= TOL, (l2, mx)
        for tuning in (0, 2001, 3000):
            f.set_tuning(tuning)
            f.reset()
            cuts = [0, 1, 2, d, d + 1, 2688, 2689, 3841, 3842 + d, 50_001, 100_003, n]
            parts = [f.process(x[2 * a:2 * b]) for a, b in zip(cuts[:-1], cuts[1:])]
            yp = np.concatenate(parts)
            assert yp.shape == ref.shape
            l2, mx = oracle.err_metrics(yp, ref)
            assert l2 <= TOL and mx <= TOL, (tuning, l2, mx)
        f.set_tuning(0)
        f.reset()
        f.set_nco(0.137)
        ref_nco = oracle.fir_nco_f64(taps, x, d, oracle.nco_phase_word(0.137))
        l2, mx = oracle.err_metrics(f.process(x), ref_nco)
        assert l2 <= TOL and mx <= TOL, ("nco", l2, mx)
        f.reset()       # the NCO through ragged pieces too (decimation 8 / 16: the one-channel filter-bank route with its
        cuts = [0, 3, d + 1, 3842 + d, 50_001, n]   # output rotation split into block, lane and slot factors)
        yp = np.concatenate([f.process(x[2 * a:2 * b]) for a, b in zip(cuts[:-1], cuts[1:])])
        l2, mx = oracle.err_metrics(yp, ref_nco)
        assert l2 <= TOL and mx <= TOL, ("nco ragged", l2, mx)
    xi = np.clip(np.round(x * 6000.0), -32768, 32767).astype(np.int16)
    xf = xi.astype(np.float32) * np.float32(2.0 ** -15)
    with fir.IfFir(taps, d, n) as f:
        f.set_input_format(fir.INPUT_I16)
        f.set_backend(fir.BACKEND_HIP_FFT)
        parts = [f.process(xi[2 * a:2 * b]) for a, b in ((0, 33_333), (33_333, n))]
        l2, mx = oracle.err_metrics(np.concatenate(parts), oracle.fir_f64(taps, xf, d))
        assert l2 <= TOL and mx <= TOL, ("i16", l2, mx)


@pytest.mark.parametrize("t,d", [(255, 3), (255, 9), (767, 3), (511, 15)])
def test_odd_decimation_kernel_blocks_queue_and_split_invariance(fir, oracle, torch_cuda, t, d):
    """Round 4: the odd-decimation kernel at sizes that run every stage of its block queue (1.05 M samples on one
    and three workgroups: bit-identical to the default launch), streams cut at multiples of its block advance (bit-identical to
    the unsplit call, as for the other tails), complex taps, and the multi-channel front's chunks on its block grid."""
    torch = torch_cuda
    n = 1_050_007
    taps = fir.bpf_design(t)
    x = oracle.synth_iq(n, 31)
    ref = oracle.fir_f64(taps, x, d)
    need = (t - 1 + 2 + 2) // 3
    adv = 3 * (1024 - 64 * (2 if need <= 128 else 4))
    with fir.IfFir(taps, d, n, dev=True) as f:
        assert f.get_backend() == fir.BACKEND_HIP_FFT
        y0 = f.process(x)
        l2, mx = oracle.err_metrics(y0, ref)
        assert l2 <= TOL and mx <= TOL, (l2, mx)
        for k in (1, 3):
            f.reset()
            f.set_tuning(2000 + k)
            assert np.array_equal(f.process(x), y0), k
        f.set_tuning(0)
        f.reset()
        unit = int(np.lcm(adv, d))
        cuts = [0, 5 * unit, 5 * unit + 40 * unit, n]
        parts = [f.process(x[2 * a:2 * b]) for a, b in zip(cuts[:-1], cuts[1:])]
        assert np.array_equal(np.concatenate(parts), y0)
        assert f.debug_queue_faults() == 0
    g = fir.bpf_design_complex(t, 0.2, 0.1)
    with fir.IfFir(g, d, 200_001, complex_taps=True) as f:
        xs = x[:2 * 200_001]
        l2, mx = oracle.err_metrics(f.process(xs), oracle.fir_ctaps_f64(g, xs, d))
        assert l2 <= TOL and mx <= TOL, ("complex taps", l2, mx)
    # the multi-channel front cuts its calls on this kernel's block grid: chunked = unchunked bit for bit, off-phase second call
    tp = np.stack([fir.bpf_design(t, 0.15, 0.25), fir.bpf_design(t, 0.02, 0.08)])
    dev_in = [torch.from_numpy(oracle.synth_iq(n, 70 + c)).cuda() for c in range(2)]
    res = {}
    with fir.IfFirMc(tp, d, n) as mc:
        for chunk in (fir.MC_NEVER_SPLIT, fir.MC_CHUNK_UNIT):
            mc.set_chunk_samples(chunk)
            eff, unit = mc.get_chunk_samples()
            assert unit == np.lcm(adv, 2 * d)
            mc.reset()
            parts = [[], []]
            for a, b in ((0, 600_001), (600_001, n)):
                m_exp = oracle.out_count(a, b - a, d)
                outs = [torch.zeros(2 * m_exp, dtype=torch.float32, device="cuda") for _ in range(2)]
                pieces = [xx[2 * a:2 * b].clone() for xx in dev_in]
                assert mc.process_device([p.data_ptr() for p in pieces], [o.data_ptr() for o in outs], b - a) == m_exp
                for c in range(2):
                    parts[c].append(outs[c].cpu().numpy())
            res[chunk] = [np.concatenate(p) for p in parts]
    for c in range(2):
        assert np.array_equal(res[fir.MC_NEVER_SPLIT][c], res[fir.MC_CHUNK_UNIT][c]), c
    l2, mx = oracle.err_metrics(res[fir.MC_CHUNK_UNIT][1], oracle.fir_f64(tp[1], dev_in[1].cpu().numpy(), d))
    assert l2 <= TOL and mx <= TOL, (l2, mx)


@pytest.mark.parametrize("i16,nco", [(False, 0.0), (True, 0.0), (False, -0.21), (True, 0.137)])
def test_multiples_of_4_are_the_decimate_by_4_outputs_thinned(fir, oracle, i16, nco):
    """Decimation 8, 12, ..., 64 run behind the decimate-by-4 tail and keep every (D/4)-th of its outputs (round 3; 6, 10, ..., 62
    likewise behind the decimate-by-2 tail): from the
    start of a stream they are, bit for bit, every (D/4)-th output of the decimate-by-4 context -- a whole-array property
    that needs no oracle -- and within SPEC tolerance of the float64 oracle like everything else."""
    n = 777_777
    x = oracle.synth_iq(n, 31)
    taps = fir.bpf_design(511, 0.0, 0.02)
    if i16:
        xin = np.clip(np.round(x * 7000.0), -32768, 32767).astype(np.int16)
        xf = xin.astype(np.float32) * np.float32(2.0 ** -15)
    else:
        xin, xf = x, x

    def run(d):
        with fir.IfFir(taps, d, n) as f:
            if i16:
                f.set_input_format(fir.INPUT_I16)
            if nco:
                f.set_nco(nco)
            assert f.get_backend() == fir.BACKEND_HIP_FFT
            return f.process(xin)
    y4 = run(4).reshape(-1, 2)
    for d in range(8, 65, 4):
        yd = run(d).reshape(-1, 2)
        assert yd.shape[0] == oracle.out_count(0, n, d)
        assert np.array_equal(yd, y4[::d // 4][:yd.shape[0]]), d
    y2 = run(2).reshape(-1, 2)      # 6, 10, ..., 62: every (D/2)-th output of the decimate-by-2 tail
    for d in range(6, 63, 4):
        yd = run(d).reshape(-1, 2)
        assert yd.shape[0] == oracle.out_count(0, n, d)
        assert np.array_equal(yd, y2[::d // 2][:yd.shape[0]]), d
    ref = oracle.fir_nco_f64(taps, xf, 20, oracle.nco_phase_word(nco)) if nco else oracle.fir_f64(taps, xf, 20)
    l2, mx = oracle.err_metrics(run(20), ref)
    assert l2 <= TOL and mx <= TOL, (l2, mx)


def test_uniform_filter_bank_int16_input(fir, oracle, torch_cuda):
    """The filter bank straight on an int16 SDR stream (both halves of §8f-1 and §8f-2 together)."""
    torch = torch_cuda
    taps = fir.bpf_design(255, 0.0, 0.03)
    n = 200_003
    xi = np.clip(np.round(oracle.synth_iq(n, 43) * 12000.0), -32768, 32767).astype(np.int16)
    xf = xi.astype(np.float32) * np.float32(2.0 ** -15)
    xd = torch.from_numpy(xi).cuda()
    slots = [1, 6, 11, 12]
    refs = [oracle.fir_nco_f64(taps, xf, 4, (s << 28) & 0xFFFFFFFF) for s in slots]
    cuts = [0, 4, 50_002, n]          # int16 pieces must start on a 16-byte boundary: multiples of 4 samples
    with fir.IfFir(taps, 4, 0) as f:
        f.set_input_format(fir.INPUT_I16)
        assert f.get_backend() == fir.BACKEND_HIP_FFT
        parts = [[] for _ in slots]
        for a, b in zip(cuts[:-1], cuts[1:]):
            m_exp = oracle.out_count(a, b - a, 4)
            outs = [torch.empty(2 * m_exp + 8, dtype=torch.float32, device="cuda") for _ in slots]
            piece = xd[2 * a:2 * b].clone()
            torch.cuda.synchronize()
            assert f.channelizer_process_device(slots, piece.data_ptr(), [o.data_ptr() for o in outs], b - a) == m_exp
            f.synchronize()
            for c in range(len(slots)):
                parts[c].append(outs[c].cpu().numpy()[:2 * m_exp])
        for c, s in enumerate(slots):
            l2, mx = oracle.err_metrics(np.concatenate(parts[c]), refs[c])
            assert l2 <= TOL and mx <= TOL, (s, l2, mx)


@pytest.mark.parametrize("t,d", [(2047, 1), (2047, 4), (2049, 8), (3073, 1), (3071, 4), (1027, 3)])
def test_fft_backend_long_filters(fir, oracle, t, d):
    """More than 1025 taps on the overlap-save backend: 32 or 48 of the 64 rows of a block are overlap (half / three
    quarters of the transform is redundant, still two orders of magnitude ahead of the direct-form kernels).  One call
    and ragged pieces against the float64 oracle; complex taps; the run queue on one workgroup."""
    rng = np.random.default_rng(t + d)
    taps = fir.bpf_design(t)
    n = 70_001
    x = np.concatenate([oracle.synth_iq(n // 2, 29), rng.standard_normal(2 * (n - n // 2)).astype(np.float32)])
    ref = oracle.fir_f64(taps, x, d)
    with fir.IfFir(taps, d, n, dev=True) as f:
        assert f.get_backend() == fir.BACKEND_HIP_FFT
        for tuning in (0, 2001):
            f.set_tuning(tuning)
            f.reset()
            l2, mx = oracle.err_metrics(f.process(x), ref)
            assert l2 <= TOL and mx <= TOL, (tuning, l2, mx)
        f.reset()
        cuts = [0, 3, 1000, 1025, 2048, 2049, 5000, 30_001, n]
        parts = [f.process(x[2 * a:2 * b]) for a, b in zip(cuts[:-1], cuts[1:])]
        l2, mx = oracle.err_metrics(np.concatenate(parts), ref)
        assert l2 <= TOL and mx <= TOL, (l2, mx)
    g = (rng.standard_normal(2 * t) / np.sqrt(t)).astype(np.float32)
    with fir.IfFir(g, d, n, complex_taps=True) as f:
        l2, mx = oracle.err_metrics(f.process(x), oracle.fir_ctaps_f64(g, x, d))
        assert l2 <= TOL and mx <= TOL, ("ctaps", l2, mx)


@pytest.mark.parametrize("t,d", [(3075, 1), (3075, 4), (3075, 8), (4095, 1), (4095, 4), (4095, 8), (4096, 1), (4096, 4),
                                 (4096, 8), (3074, 3), (4095, 2), (4096, 6), (3075, 12), (4095, 64), (3500, 10)])
def test_fft_backend_two_partitions(fir, oracle, t, d):
    """3074..4096 taps on the overlap-save backend: h = (h_a, h_b) with 2048 taps in h_a; one launch
    computes h_a * x, a second one adds h_b * x(n - 2048) (the same 32-row kernel reading the input 2048 samples late,
    accumulating store; even decimations behind the decimating tails since round 3, the second partition adding to the
    decimated outputs of the first).  Against the float64 oracle: one call, ragged pieces shorter and longer than the delay, a
    small grid, int16 input, complex taps, the NCO; and against the tap-split kernel."""
    rng = np.random.default_rng(t + d)
    def design(lo, hi):                                  # the designer makes odd lengths: an even one gets a small last tap
        if t % 2:
            return fir.bpf_design(t, lo, hi)
        return np.concatenate([fir.bpf_design(t - 1, lo, hi), np.array([0.001], np.float32)]).astype(np.float32)

    taps = design(0.15, 0.25)
    n = 90_001
    x = np.concatenate([oracle.synth_iq(n // 2, 33), rng.standard_normal(2 * (n - n // 2)).astype(np.float32)])
    ref = oracle.fir_f64(taps, x, d)
    with fir.IfFir(taps, d, n, dev=True) as f:
        assert f.get_backend() == fir.BACKEND_HIP_FFT
        for tuning in (0, 2001):
            f.set_tuning(tuning)
            f.reset()
            y = f.process(x)
            l2, mx = oracle.err_metrics(y, ref)
            assert l2 <= TOL and mx <= TOL, (tuning, l2, mx)
        f.set_tuning(0)
        f.reset()
        cuts = [0, 3, 1000, 2047, 2048, 2049, 4100, 6200, 30_001, 30_002, 70_000, n]
        parts = [f.process(x[2 * a:2 * b]) for a, b in zip(cuts[:-1], cuts[1:])]
        l2, mx = oracle.err_metrics(np.concatenate(parts), ref)
        assert l2 <= TOL and mx <= TOL, ("pieces", l2, mx)
        f.set_backend(fir.BACKEND_HIP_TAPSPLIT)
        f.reset()
        scale = float(np.abs(ref).max())
        assert float(np.abs(f.process(x) - y).max()) <= 2e-6 * scale
    if d in (1, 4):
        xi = np.clip(np.round(x * 8000.0), -32768, 32767).astype(np.int16)
        with fir.IfFir(taps, d, n) as f:
            f.set_input_format(fir.INPUT_I16)
            parts = [f.process(xi[:2 * 40_001]), f.process(xi[2 * 40_001:])]
            l2, mx = oracle.err_metrics(np.concatenate(parts), oracle.fir_f64(taps, xi.astype(np.float32) * np.float32(2.0 ** -15), d))
            assert l2 <= TOL and mx <= TOL, ("i16", l2, mx)
        g = (rng.standard_normal(2 * t) / np.sqrt(t)).astype(np.float32)
        with fir.IfFir(g, d, n, complex_taps=True) as f:
            l2, mx = oracle.err_metrics(f.process(x), oracle.fir_ctaps_f64(g, x, d))
            assert l2 <= TOL and mx <= TOL, ("ctaps", l2, mx)
        lp = design(0.0, 0.05)
        with fir.IfFir(lp, d, n) as f:
            f.set_nco(0.2003)
            parts = [f.process(x[:2 * 50_003]), f.process(x[2 * 50_003:])]
            l2, mx = oracle.err_metrics(np.concatenate(parts), oracle.fir_nco_f64(lp, x, d, oracle.nco_phase_word(0.2003)))
            assert l2 <= TOL and mx <= TOL, ("nco", l2, mx)


@pytest.mark.parametrize("d,i16", [(4, False), (3, False), (4, True), (1, False)])
def test_host_path_pipelined_chunks(fir, oracle, d, i16):
    """if_fir_process on a long host buffer: the input goes through the device in chunks on three streams (copy in,
    kernels, copy out).  Same numbers as the chunk-free device path and the oracle (windows), with pageable and with
    page-locked (if_fir_host_alloc) buffers, for decimations that do and do not divide the chunk grid."""
    n = 3 * (1 << 22) + 12_347
    taps = fir.bpf_design(255)
    x = oracle.synth_iq(n, 61)
    if i16:
        src = np.clip(np.round(x * 12000.0), -32768, 32767).astype(np.int16)
        x = src.astype(np.float32) * np.float32(2.0 ** -15)
    else:
        src = x
    m = oracle.out_count(0, n, d)
    with fir.IfFir(taps, d, n) as f:
        if i16:
            f.set_input_format(fir.INPUT_I16)
        y_pageable = f.process(src)
        assert y_pageable.size == 2 * m
        # windows against the oracle: start, a chunk seam, the end
        for start in (0, ((1 << 22) // (4 * d)) * 4 * d - 600, n - 5000):
            start -= start % d
            w = min(4000, n - start)
            lo = max(0, start - 254)
            hist = np.zeros(2 * 254, np.float32)
            hist[2 * (254 - (start - lo)):] = x[2 * lo:2 * start]
            ref = oracle.fir_f64(taps, x[2 * start:2 * (start + w)], d, hist, start)
            got = y_pageable[2 * (start // d):2 * (start // d) + ref.size]
            l2, mx = oracle.err_metrics(got, ref)
            assert l2 <= TOL and mx <= TOL, (start, l2, mx)
        # page-locked buffers: identical result
        f.reset()
        xin = f.host_alloc(2 * n, np.int16 if i16 else np.float32)
        yout = f.host_alloc(2 * m + 4, np.float32)
        xin[:] = src
        yout[:] = 7.0
        assert f.process_into(xin, yout) == m
        assert np.array_equal(yout[:2 * m], y_pageable) and np.all(yout[2 * m:] == 7.0)
        f.host_free(xin)
        f.host_free(yout)
        # and a second long call continues the stream (history and phase survive the chunking)
        f.reset()
        y2 = np.concatenate([f.process(src[:2 * (n // 2 + 1)]), f.process(src[2 * (n // 2 + 1):])])
        scale = np.max(np.abs(y_pageable))
        assert y2.shape == y_pageable.shape and np.max(np.abs(y2 - y_pageable)) <= 2e-6 * scale


@pytest.mark.filterwarnings("ignore:The CUDA Graph is empty")
def test_stream_capture_is_refused(fir, oracle, torch_cuda):
    """A call's launch arguments carry host-side streaming state (sample index, phase, history ping-pong, run-queue
    base): replaying them from a hipGraph would be wrong, so a capturing stream is refused with a message."""
    torch = torch_cuda
    n = 8192
    x = torch.from_numpy(oracle.synth_iq(n, 2)).cuda()
    with fir.IfFir(fir.bpf_design(255), 4, n) as f:
        y = torch.empty(2 * f.out_count(n), dtype=torch.float32, device="cuda")
        s = torch.cuda.Stream()
        f.set_stream(s.cuda_stream)
        f.process_device(x.data_ptr(), y.data_ptr(), n)      # warm: attributes, tables
        f.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            g.capture_begin()
            try:
                with pytest.raises(fir.IfFirError, match="captured"):
                    f.process_device(x.data_ptr(), y.data_ptr(), n)
            finally:
                g.capture_end()
        f.reset()
        f.process_device(x.data_ptr(), y.data_ptr(), n)      # the context still works
        f.synchronize()
        l2, mx = oracle.err_metrics(y.cpu().numpy(), oracle.fir_f64(fir.bpf_design(255), x.cpu().numpy(), 4))
        assert l2 <= TOL and mx <= TOL


def test_block_queue_fault_is_reported_to_the_caller(fir, oracle, torch_cuda):
    """The overlap-save kernel's block queue bounds every wait; a wave whose wait expires leaves its blocks unwritten and counts
    a fault.  That must never happen -- and if it does the caller has to hear about it.  Development launch 512 makes the
    waves of workgroup 0 do exactly that (count a fault, leave): the launch ends with outputs missing, if_fir_synchronize
    fails with the queue's message, and the same context filters correctly again afterwards.  (The bounded waits
    themselves run in the host simulation of the queue code, tests/c/fft_queue_sim.cpp.)"""
    torch = torch_cuda
    n = 1 << 25
    taps = fir.bpf_design(255)
    x = torch.empty(2 * n, dtype=torch.float32, device="cuda")
    old = os.environ.get("IF_FIR_DEBUG")
    os.environ["IF_FIR_DEBUG"] = "1"
    try:
        with fir.IfFir(taps, 4, 0, dev=True) as f:
            m = f.out_count(n)
            good = torch.zeros(2 * m, dtype=torch.float32, device="cuda")
            bad = torch.zeros(2 * m, dtype=torch.float32, device="cuda")
            f.synth_device(x.data_ptr(), 0, n, 0)
            assert f.process_device(x.data_ptr(), good.data_ptr(), n) == m
            f.synchronize()
            assert f.debug_queue_faults() == 0
            f.reset()
            f.set_tuning(1000000 + 512)
            assert f.process_device(x.data_ptr(), bad.data_ptr(), n) == m     # asynchronous: the launch itself is accepted
            with pytest.raises(fir.IfFirError, match="bounded wait"):
                f.synchronize()
            assert not torch.equal(good, bad) and int((bad.view(-1, 2).abs().sum(1) == 0).sum()) >= 960   # blocks left unwritten
            f.set_tuning(0)
            f.reset()
            again = torch.zeros(2 * m, dtype=torch.float32, device="cuda")
            assert f.process_device(x.data_ptr(), again.data_ptr(), n) == m
            f.synchronize()
            assert torch.equal(good, again)
    finally:
        if old is None:
            os.environ.pop("IF_FIR_DEBUG", None)
        else:
            os.environ["IF_FIR_DEBUG"] = old


def test_block_queue_counters_alternate_over_many_launches(fir, oracle):
    """Overlap-save launches draw block groups from one of two global counters and zero the other one for the launch
    behind them.  Many back-to-back calls on small grids (most blocks then come through the global counter) give the
    same numbers as a fresh context that sees the stream in one call, also after another backend used the queue words."""
    n = 700_001
    taps = fir.bpf_design(255)
    x = oracle.synth_iq(n, 71)
    cuts = [0, 100_000, 250_000, 250_004, 400_001, 500_003, 650_000, n]
    with fir.IfFir(taps, 4, n, dev=True) as f:
        assert f.get_backend() == fir.BACKEND_HIP_FFT
        # (the same cuts everywhere: where a block starts decides the last bits of an FFT result)
        ref = np.concatenate([f.process(x[2 * a:2 * b]) for a, b in zip(cuts[:-1], cuts[1:])])
        for grid in (2001, 2002, 2005, 0):
            f.reset()
            f.set_tuning(grid)
            parts = [f.process(x[2 * a:2 * b]) for a, b in zip(cuts[:-1], cuts[1:])]
            assert np.array_equal(np.concatenate(parts), ref), grid
        f.set_tuning(2003)
        f.reset()
        first = f.process(x[:2 * 300_000])
        f.set_backend(fir.BACKEND_HIP_DIRECT)      # the direct kernel re-zeroes and uses the same queue words
        f.process(x[:2 * 100_000])
        f.set_backend(fir.BACKEND_HIP_FFT)
        f.reset()
        again = np.concatenate([f.process(x[:2 * 300_000]), f.process(x[2 * 300_000:])])
        assert np.array_equal(again[:first.size], first)
        l2, mx = oracle.err_metrics(again, oracle.fir_f64(taps, x, 4))
        assert l2 <= TOL and mx <= TOL
    l2, mx = oracle.err_metrics(ref, oracle.fir_f64(taps, x, 4))
    assert l2 <= TOL and mx <= TOL


def test_multi_channel_front_int16_and_reset(fir, oracle, torch_cuda):
    """if_fir_mc_set_input_format / if_fir_mc_reset on one rank: two int16 channels, then a reset and the same stream
    again give the same numbers."""
    torch = torch_cuda
    n, d = 40_004, 4
    taps = np.stack([fir.bpf_design(255, 0.1, 0.2), fir.bpf_design(255, 0.25, 0.4)])
    xi = [np.clip(np.round(oracle.synth_iq(n, 80 + c) * 9000.0), -32768, 32767).astype(np.int16) for c in range(2)]
    dev_in = [torch.from_numpy(v).cuda() for v in xi]
    m = oracle.out_count(0, n, d)
    with fir.IfFirMc(taps, d, n) as mc:
        mc.set_input_format(fir.INPUT_I16)
        runs = []
        for _ in range(2):
            outs = [torch.zeros(2 * m, dtype=torch.float32, device="cuda") for _ in range(2)]
            assert mc.process_device([p.data_ptr() for p in dev_in], [o.data_ptr() for o in outs], n) == m
            runs.append([o.cpu().numpy() for o in outs])
            mc.reset()
        for c in range(2):
            assert np.array_equal(runs[0][c], runs[1][c])
            ref = oracle.fir_f64(taps[c], xi[c].astype(np.float32) * np.float32(2.0 ** -15), d)
            l2, mx = oracle.err_metrics(runs[0][c], ref)
            assert l2 <= TOL and mx <= TOL, (c, l2, mx)
        with pytest.raises(fir.IfFirError):
            mc.set_input_format(7)


@pytest.mark.parametrize("n", [1, 2, 3, 255, 256, 100_001, 8_388_609])
def test_in_band_power_measurement(fir, oracle, torch_cuda, n):
    """if_fir_power_device (the measurement half of a level-control loop): mean |y|^2 of a device buffer,
    accumulated in float64, against numpy's float64 mean; and on a filtered stream: the stop-band tone of the synthetic
    input is gone, the in-band tone keeps its power."""
    torch = torch_cuda
    x = oracle.synth_iq(n, 91)
    xd = torch.from_numpy(x).cuda()
    with fir.IfFir(fir.bpf_design(255), 4, 0) as f:
        got = f.power_device(xd.data_ptr(), n)
        want = float(np.mean(np.sum(x.astype(np.float64).reshape(-1, 2) ** 2, axis=1)))
        assert abs(got - want) <= 1e-12 * want, (got, want)
        assert f.power_device(xd.data_ptr(), 0) == 0.0
        with pytest.raises(fir.IfFirError):
            f.power_device(xd.data_ptr() + 8, 1)          # misaligned
        if n > 100_000:
            y = torch.empty(2 * f.out_count(n), dtype=torch.float32, device="cuda")
            m = f.process_device(xd.data_ptr(), y.data_ptr(), n)
            f.synchronize()
            p_out = f.power_device(y.data_ptr(), m)
            # input: two tones of power 0.25 each + uniform noise (variance 2 * 0.25^2 / 12 * ... small); the 0.15-0.25
            # band-pass keeps the 0.20 tone (gain 1) and a tenth of the noise
            assert 0.25 < p_out < 0.26 and want > 0.5
