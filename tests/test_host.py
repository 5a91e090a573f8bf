"""CPU tests of the host side: the C-ABI library loads and exports every symbol include/if_fir.h declares, the C tap
designer matches the oracle and the golden taps, and the library fails loudly (no CPU fallback) without a GPU."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "if_fir_golden.npz")


def test_library_exports_every_declared_symbol(fir):
    header = open(os.path.join(ROOT, "include", "if_fir.h")).read()
    declared = set(re.findall(r"\b(if_(?:fir|bpf)_[a-z_]+)\s*\(", header))
    assert declared == set(fir.EXPORTS), declared ^ set(fir.EXPORTS)
    lib = fir.lib()
    for name in declared:
        assert hasattr(lib, name), name
    # the development hooks (include/if_fir_debug.h) exist in libif_fir_dev.so only: the product exports none of them
    import subprocess
    dbg = open(os.path.join(ROOT, "include", "if_fir_debug.h")).read()
    dev_declared = set(re.findall(r"^\w+ \*?(if_fir_[a-z_]+)\s*\(", dbg, re.M))
    assert dev_declared == set(fir.DEV_EXPORTS), dev_declared ^ set(fir.DEV_EXPORTS)
    product = subprocess.run(["nm", "-D", "--defined-only", fir.LIB_PATH], capture_output=True, text=True).stdout
    product_syms = {line.split()[-1] for line in product.splitlines() if line.strip()}
    assert not [n for n in product_syms if "debug" in n or n in dev_declared], product_syms & dev_declared
    assert {n for n in product_syms if n.startswith(("if_fir_", "if_bpf_"))} == declared
    dev = fir.dev_lib()
    for name in declared | dev_declared:
        assert hasattr(dev, name), name
    # include/wb_detect.h lives in the same library
    import __graft_entry__ as g
    wb = g.load_pkg().wb_detect
    header = open(os.path.join(ROOT, "include", "wb_detect.h")).read()
    declared = set(re.findall(r"\b(wb_detect_[a-z_]+)\s*\(", header))
    assert declared == set(wb.EXPORTS), declared ^ set(wb.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name


def test_library_does_not_link_the_oracle_or_torch():
    import subprocess
    so = os.path.join(ROOT, "qo-100-tools_amd", "libif_fir.so")
    needed = subprocess.run(["readelf", "-d", so], capture_output=True, text=True).stdout
    assert "oracle" not in needed and "torch" not in needed and "amdhip64" in needed
    syms = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True).stdout
    assert "oracle_" not in syms


@pytest.mark.parametrize("t", [127, 255, 1023])
def test_c_designer_matches_oracle_and_golden(fir, oracle, t):
    h = fir.bpf_design(t)
    assert np.array_equal(h, oracle.bpf_design(t))
    assert np.array_equal(h, np.load(GOLD)["taps_%d" % t])
    for win, name in [(fir.WINDOW_RECT, "rect"), (fir.WINDOW_HAMMING, "hamming"), (fir.WINDOW_HANN, "hann")]:
        assert np.array_equal(fir.bpf_design(t, 0.1, 0.3, win), oracle.bpf_design(t, 0.1, 0.3, name))


def test_c_complex_designer_matches_oracle_and_golden(fir, oracle):
    g = fir.bpf_design_complex(255, 0.2, 0.1)
    assert np.array_equal(g, oracle.bpf_design_complex(255, 0.2, 0.1))
    assert np.array_equal(g, np.load(GOLD)["ctaps_255"])
    assert np.array_equal(fir.bpf_design_complex(127, -0.3, 0.05, fir.WINDOW_HAMMING),
                          oracle.bpf_design_complex(127, -0.3, 0.05, "hamming"))
    for args in [(128, 0.2, 0.1), (255, 0.6, 0.1), (255, 0.2, 0.0), (255, 0.2, 1.5)]:
        with pytest.raises(fir.IfFirError):
            fir.bpf_design_complex(*args)


def test_c_designer_rejects_bad_arguments(fir):
    for args in [(128,), (1,), (0,), (4097,), (127, 0.3, 0.2), (127, 0.1, 0.6), (127, -0.1, 0.2), (127, 0.1, 0.2, 9)]:
        with pytest.raises(fir.IfFirError):
            fir.bpf_design(*args)


def test_init_argument_errors_and_no_cpu_fallback(fir):
    import torch
    taps = fir.bpf_design(127)
    for bad in [dict(taps=np.zeros(0, np.float32)), dict(taps=np.zeros(5000, np.float32)),
                dict(taps=taps, decimation=0), dict(taps=taps, decimation=65),
                dict(taps=np.array([1.0, np.nan], np.float32))]:
        with pytest.raises(fir.IfFirError):
            fir.IfFir(**bad)
    if not torch.cuda.is_available():
        with pytest.raises(fir.IfFirError, match="no HIP device"):
            fir.IfFir(taps)   # must fail loudly, never fall back to a CPU path


def test_generated_walk_header_is_current():
    """csrc/generated/if_fir_walk_gen.h is the committed output of tools/gen_walk.py (regenerating changes nothing)."""
    import subprocess
    import tempfile
    out = os.path.join(tempfile.mkdtemp(), "walk.h")
    subprocess.check_call(["python3", os.path.join(ROOT, "tools", "gen_walk.py"), "--out", out])
    committed = open(os.path.join(ROOT, "qo-100-tools_amd", "csrc", "generated", "if_fir_walk_gen.h")).read()
    assert open(out).read() == committed


def test_mc_owner_matches_channel_map_and_init_argument_errors(fir):
    """The multi-channel C front (if_fir_mc_*): ownership rule = channel_shard.channel_map, and the argument checks that
    need no GPU fail with a message instead of touching a device."""
    import __graft_entry__ as g
    cs = g.load_pkg().channel_shard
    for world in (1, 2, 3, 8):
        cmap = cs.channel_map(8, world)
        for rank, chans in enumerate(cmap):
            for c in chans:
                assert fir.mc_owner(c, world) == rank
    taps = np.stack([fir.bpf_design(31), fir.bpf_design(31, 0.05, 0.1)])
    with pytest.raises(fir.IfFirError, match="invalid argument"):
        fir.IfFirMc(taps, 1, 0)                                   # max samples must be > 0
    with pytest.raises(fir.IfFirError, match="invalid argument"):
        fir.IfFirMc(taps, 1, 1000, rank=2, world=2)               # rank out of range
    with pytest.raises(fir.IfFirError, match="unique id"):
        fir.IfFirMc(taps, 1, 1000, rank=0, world=2)               # two ranks without the bootstrap id


def _tsw(m):
    """position of entry m in an array of the full-rate pipeline's shared twiddle table (tsw in csrc/if_fir_fft.hip)"""
    return ((m & ~31) | ((m ^ (m >> 5)) & 1) | ((((m >> 1) ^ (m >> 6)) & 1) << 1) | (m & 4) | ((((m >> 3) ^ (m >> 7)) & 1) << 3) | (m & 16))


def test_shared_twiddle_table_gathers_are_conflict_free():
    """Round 4: the full-rate pipeline's last inverse pass reads the shared table at m = lane + 64 rho + 256 q (second stage) and
    m = 4 lane + 256 rho (first stage), its forward pass 3 at m = 4 (lane / 16) + i + 16 (lane % 16) + 256 q.  With the position
    function tsw the 32 lanes of either half of a wave always land on 32 different 8-byte bank slots, and + 256 q stays a plain offset."""
    lane = np.arange(64)
    pats = [lane + 64 * rho + 256 * q for rho in range(4) for q in range(4)]
    pats += [4 * lane + 256 * rho for rho in range(4)]
    pats += [4 * (lane // 16) + i + 16 * (lane % 16) + 256 * q for i in range(4) for q in range(4)]
    for ms in pats:
        for half in (0, 1):
            slots = {_tsw(int(v)) % 32 for v in ms[32 * half:32 * half + 32]}
            assert len(slots) == 32, ms
    assert all(_tsw(m + 256) == _tsw(m) + 256 for m in range(768))
    assert all(_tsw(4 * (g + 4 * k1) + i) == _tsw(4 * (g + 4 * k1)) ^ i for g in range(4) for k1 in range(16) for i in range(4))


def _bin_of(i, k2, lane):
    """frequency bin held by table entry (i, k2, lane) of the overlap-save kernel (csrc/if_fir_fft.hip, fft_build_tables)"""
    return (4 * (lane // 16) + i) + 16 * (lane % 16) + 256 * k2


@pytest.mark.parametrize("t,d,ctaps", [(255, 1, False), (255, 4, False), (1023, 4, False), (63, 3, False), (127, 4, True),
                                       (2047, 8, False)])
def test_overlap_save_tables_against_numpy(fir, t, d, ctaps):
    """Host-side table builder (float64 math in C++, no GPU involved) against an independent numpy derivation: the
    twiddle tables, H = FFT(taps)/4096 in the kernel's lane order, and for decimate-by-4 the merged table G, which must
    reproduce 'second radix-4 stage, multiply by H, fold the four aliases' exactly as a linear map."""
    rng = np.random.default_rng(t + d)
    if ctaps:
        h = (rng.standard_normal(t) + 1j * rng.standard_normal(t)) / np.sqrt(t)
        taps = np.ascontiguousarray(h.astype(np.complex64)).view(np.float32)
        h = taps.view(np.complex64).astype(np.complex128)
    else:
        taps = (rng.standard_normal(t) / np.sqrt(t)).astype(np.float32)
        h = taps.astype(np.complex128)
    nco_delta = 0x12345678
    tab = fir.debug_fft_tables(taps, d, complex_taps=ctaps, nco_delta=nco_delta)
    lane = np.arange(64)
    W = lambda n, e: np.exp(-2j * np.pi * (np.asarray(e) % n) / n)   # noqa: E731
    # twiddles
    merged = d % 4 == 0          # decimation 4 and its multiples: the decimate-by-4 image (merged table G')
    full_rate = d % 2 == 1       # D = 1 and the selecting store: the full-rate pipeline's image (round 4)

    def triple(e1, e2, e3):      # (c, t) entries of one butterfly -> its three twiddles; the third entry holds (c3 / c1, t3)
        c1, t1 = e1.real.astype(np.float64), e1.imag.astype(np.float64)
        c2, t2 = e2.real.astype(np.float64), e2.imag.astype(np.float64)
        r3, t3 = e3.real.astype(np.float64), e3.imag.astype(np.float64)
        return c1 * (1 + 1j * t1), c2 * (1 + 1j * t2), r3 * c1 * (1 + 1j * t3)

    def check_entries(entries, base):           # entries: (15, n) complex64 = (c, t) pairs; base: (n,) complex
        w = triple(entries[0], entries[1], entries[2])
        for k, got in zip((4, 8, 12), w):
            assert np.allclose(got, base ** k, atol=3e-7), k
        for q in range(4):
            b = base * W(16, q)
            w = triple(entries[3 + 3 * q], entries[4 + 3 * q], entries[5 + 3 * q])
            for k, got in zip((1, 2, 3), w):
                assert np.allclose(got, b ** k, atol=3e-7), (q, k)
        assert np.all(np.isfinite(entries.view(np.float32)))

    if not merged and not full_rate:
        for rho in range(4):
            for k0 in range(16):
                assert np.allclose(tab["tw1"][(rho * 16 + k0) * 64:(rho * 16 + k0) * 64 + 64], W(4096, (lane + 64 * rho) * k0), atol=1e-7)
        for k1 in range(16):
            assert np.allclose(tab["tw2"][k1 * 16:k1 * 16 + 16], W(256, np.arange(16) * k1), atol=1e-7)
    elif full_rate:
        # forward pass 2 and the first stage of forward pass 3 as in the decimate-by-4 image; the shared table T of the triples
        # (b, b^2, b^3), b = W4096^m, at position tsw(m) of three arrays of 1024 entries; inverse pass 2: b = W256^(lane % 16)
        g, m = lane // 16, lane % 16
        for i in range(4):
            base3 = W(4096, (4 * g + i) + 16 * m)
            e = tab["tw1"][i * 3 * 64:(i * 3 + 3) * 64].reshape(3, 64)
            for k, got in zip((4, 8, 12), triple(e[0], e[1], e[2])):
                assert np.allclose(got, base3 ** k, atol=3e-7), (i, k)
            check_entries(tab["tw2"][i * 60:(i + 1) * 60].reshape(15, 4), W(256, 4 * np.arange(4) + i))
        tt = tab["tw1"][1024:4096].reshape(3, 1024)
        mm = np.arange(1024)
        pos = np.array([_tsw(int(v)) for v in mm])
        assert sorted(pos) == list(range(1024))
        for k, got in zip((1, 2, 3), triple(tt[0][pos], tt[1][pos], tt[2][pos])):
            assert np.allclose(got, W(4096, mm) ** k, atol=3e-7), k
        check_entries(tab["twd"][:15 * 16].reshape(15, 16), W(256, np.arange(16)))
    else:
        # round 4: the decimate-by-4 kernels take their twiddles in (cos, tan) form, w = c (1 + j t) stored as (c, t), on the
        # INPUTS of the 16-point transforms: 15 entries per base twiddle b -- b^4, b^8, b^12 (the third as (c3 / c1, t3)), then for
        # q = 0..3 the three twiddles (b W16^q)^1..3 (fft16_tw in csrc/if_fir_fft.hip)
        g, m = lane // 16, lane % 16
        for i in range(4):
            base3 = W(4096, (4 * g + i) + 16 * m)       # pass 3: only the first-stage entries are kept
            e = tab["tw1"][i * 3 * 64:(i * 3 + 3) * 64].reshape(3, 64)
            w = triple(e[0], e[1], e[2])
            for k, got in zip((4, 8, 12), w):
                assert np.allclose(got, base3 ** k, atol=3e-7), (i, k)
            check_entries(tab["tw2"][i * 60:(i + 1) * 60].reshape(15, 4), W(256, 4 * np.arange(4) + i))     # pass 2: b = W256^k0
        check_entries(tab["twd"][:15 * 64].reshape(15, 64), W(1024, lane))                                  # inverse, last pass
        check_entries(tab["twe"][:15 * 4].reshape(15, 4), W(64, np.arange(4)))                              # inverse, middle pass
    r = np.arange(64, dtype=np.uint64)
    ph = ((r * np.uint64(64) * np.uint64(nco_delta)) & np.uint64(0xFFFFFFFF)).astype(np.float64) / 2.0 ** 32
    assert np.allclose(tab["ncob"], np.exp(2j * np.pi * ph), atol=1e-7)
    for i in range(4):   # decimate-by-2 inverse (round 3): W2048^(16 k1 + k0), k0 = 4 (lane // 16) + i, k1 = lane % 16
        assert np.allclose(tab["twf"][i * 64:i * 64 + 64], W(2048, 16 * (lane % 16) + 4 * (lane // 16) + i), atol=1e-7)
    # H in lane order
    H = np.fft.fft(h, 4096) / 4096.0
    Hp = np.zeros((4, 16, 64), dtype=np.complex128)
    for i in range(4):
        for k2 in range(16):
            Hp[i, k2] = H[_bin_of(i, k2, lane)]
    hp = tab["hp"].astype(np.complex128).reshape(4, 16, 64)
    scale = np.max(np.abs(H))
    if not merged:
        assert np.max(np.abs(hp - Hp)) <= 2e-7 * scale
        return
    # merged table: for random pass-2 outputs t[m] (m = time digit of the last 16-point transform),
    #   sum_p H(q + 4p) * FFT16(t)[q + 4p]  ==  sum_m0 y[q][m0] * G[m0][q],   y[q][m0] = sum_m1 t[m0 + 4 m1] W4^(m1 q)
    # (round 4: the inputs of that transform still carry b^n2, b = W4096^(k0 + 16 k1) -- the twiddles of passes 1 and 2, moved to
    # its inputs; the first stage applies b^(4 m1) inside its butterflies and the table holds G' = b^m0 G)
    G = hp                                                    # entry (i, 4*m0 + q, lane)
    tt = rng.standard_normal((4, 16, 64)) + 1j * rng.standard_normal((4, 16, 64))
    for i in range(4):
        b3 = W(4096, (4 * (lane // 16) + i) + 16 * (lane % 16))
        Y = np.fft.fft(tt[i] * b3[None, :] ** np.arange(16)[:, None], axis=0)      # over the time digit
        want = np.stack([sum(Hp[i, q + 4 * p] * Y[q + 4 * p] for p in range(4)) for q in range(4)])
        y = np.zeros((4, 4, 64), dtype=np.complex128)          # y[q][m0]
        for q in range(4):
            for m0 in range(4):
                y[q, m0] = sum(tt[i, m0 + 4 * m1] * b3 ** (4 * m1) * W(4, m1 * q) for m1 in range(4))
        got = np.stack([sum(y[q, m0] * G[i, 4 * m0 + q] for m0 in range(4)) for q in range(4)])
        assert np.max(np.abs(got - want)) <= 1e-6 * np.max(np.abs(want)), (i, np.max(np.abs(got - want)))
    # filter-bank identity: a channel at slot s uses G_s[m0][q] = W16^(m0 s) G[m0][(q - s) mod 4],
    # which must equal the merged table of the prototype shifted up by 256 s bins
    for s in (1, 6, 11):
        Hs = np.roll(H, 256 * s)
        for i in range(4):
            for m0 in range(4):
                for q in range(4):
                    b3 = W(4096, (4 * (lane // 16) + i) + 16 * (lane % 16))
                    direct = b3 ** m0 * W(16, m0 * q) * sum(Hs[_bin_of(i, q + 4 * p, lane)] * W(4, m0 * p) for p in range(4))
                    via = W(16, m0 * s) * G[i, 4 * m0 + ((q - s) % 4)]
                    assert np.max(np.abs(direct - via)) <= 3e-7 * scale, (s, i, m0, q)


@pytest.mark.parametrize("t,d,ctaps", [(255, 3, False), (383, 9, False), (767, 3, True), (511, 15, False), (3, 3, False)])
def test_odd_decimation_tables_against_numpy(fir, t, d, ctaps):
    """Round 4: the table image of the odd-decimation kernel (blocks of 3 x 1024 samples: three forward 1024-point transforms of
    the phase streams x_p[m] = x[3 m + p], Z = sum_p F_p G_p, one inverse; tools/fft_model.py odd_block).  G_p against numpy's FFT
    of the polyphase components, the twiddles as (cos, tan) entries, and -- the point -- the whole block algebra: the tables
    driven through the numpy model give y[3 m] of a random block."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fft_model as fm
    rng = np.random.default_rng(t + d)
    if ctaps:
        h = (rng.standard_normal(t) + 1j * rng.standard_normal(t)) / np.sqrt(t)
        taps = np.ascontiguousarray(h.astype(np.complex64)).view(np.float32)
        h = taps.view(np.complex64).astype(np.complex128)
    else:
        taps = (rng.standard_normal(t) / np.sqrt(t)).astype(np.float32)
        h = taps.astype(np.complex128)
    tab = fir.debug_fft_tables_odd(taps, d, complex_taps=ctaps, nco_delta=0x01234567)
    lane = np.arange(64)
    W = lambda n, e: np.exp(-2j * np.pi * (np.asarray(e) % n) / n)   # noqa: E731
    G = tab["g"].astype(np.complex128).reshape(3, 16, 64)
    scale = 0.0
    for p in range(3):
        Gp = np.fft.fft(fm.odd_phase_taps(h, 3, p)) / 1024.0
        scale = max(scale, np.max(np.abs(Gp)))
        for slot in range(16):
            assert np.max(np.abs(G[p, slot] - Gp[fm.k1024_of(slot, lane)])) <= 3e-7 * max(scale, 1e-30), (p, slot)

    def cplx(e, ref=None):                       # (c, t) -> c (1 + j t); the third entry of a butterfly holds c3 / c1
        c, tt = e.real.astype(np.float64), e.imag.astype(np.float64)
        return (c if ref is None else c * ref) * (1 + 1j * tt)

    tb = tab["tb"][:15 * 16].reshape(15, 16)     # forward middle pass: b = W256^k0
    base = W(256, np.arange(16))
    assert np.allclose(cplx(tb[0]), base ** 4, atol=3e-7) and np.allclose(cplx(tb[1]), base ** 8, atol=3e-7)
    assert np.allclose(cplx(tb[2], tb[0].real.astype(np.float64)), base ** 12, atol=3e-7)
    for q in range(4):
        b = base * W(16, q)
        assert np.allclose(cplx(tb[3 + 3 * q]), b, atol=3e-7) and np.allclose(cplx(tb[4 + 3 * q]), b ** 2, atol=3e-7)
        assert np.allclose(cplx(tb[5 + 3 * q], tb[3 + 3 * q].real.astype(np.float64)), b ** 3, atol=3e-7)
    tc = tab["tc"].reshape(4, 3, 64)             # forward last pass (4-point): b, b^2, b^3 with b = W1024^(k0 + 16 k1)
    for i in range(4):
        b = W(1024, (4 * (lane // 16) + i) + 16 * (lane % 16))
        assert np.allclose(cplx(tc[i, 0]), b, atol=3e-7) and np.allclose(cplx(tc[i, 1]), b ** 2, atol=3e-7)
        assert np.allclose(cplx(tc[i, 2], tc[i, 0].real.astype(np.float64)), b ** 3, atol=3e-7)
    r = np.arange(64, dtype=np.uint64)
    ph = ((r * np.uint64(64) * np.uint64(0x01234567)) & np.uint64(0xFFFFFFFF)).astype(np.float64) / 2.0 ** 32
    assert np.allclose(tab["ncob"], np.exp(2j * np.pi * ph), atol=1e-7)
    # the block algebra with the library's G tables in place of the model's
    xb = rng.standard_normal(3 * 1024) + 1j * rng.standard_normal(3 * 1024)
    Z = 0
    for p in range(3):
        reg = np.stack([xb[p::3][64 * rr + lane] for rr in range(16)])
        Z = Z + fm.forward_1024(reg) * G[p] * 1024.0
    y = fm.inverse_1024_from_z(Z)
    full = np.convolve(xb, h)[:3 * 1024][::3]
    ovl = -(-(t - 1 + 2) // 3)
    assert ovl <= 256 and np.max(np.abs(y[ovl:] - full[ovl:])) <= 2e-6 * max(np.max(np.abs(full)), 1e-30)


def test_odd_decimation_routing(fir):
    """Which (taps, decimation) pairs the odd-decimation kernel serves: decimation 3, 9, 15, ..., 63 (a multiple of 3) with at
    most 3 x 256 - 1 = 767 taps; everything else keeps its route (the host-only table hook refuses them)."""
    for t, d, ok in ((255, 3, True), (767, 3, True), (768, 3, False), (1535, 3, False), (255, 9, True), (255, 63, True), (255, 5, False),
                     (255, 7, False), (255, 6, False), (255, 1, False), (4095, 3, False), (1, 3, True)):
        try:
            fir.debug_fft_tables_odd(np.ones(t, np.float32), d)
            served = True
        except fir.IfFirError:
            served = False
        assert served == ok, (t, d)


def test_overlap_save_tables_refuse_unsupported(fir):
    with pytest.raises(fir.IfFirError):
        fir.debug_fft_tables(np.ones(3075, np.float32), 1)


def test_overlap_save_block_queue_hands_out_every_block_once(fir):
    """The ARITHMETIC of the two-level block queue (which global group a workgroup's local group g is; the kernel's own
    queue code runs under test_block_queue_kernel_code_under_host_simulation): a workgroup's slot s is block s % 8 of its
    local group s // 8; the first `ahead` local groups are static (global groups b, wgs + b, ...), local group g >= ahead is
    global group ahead * wgs + ticket, the ticket drawn from the launch's counter by whoever takes slot 0 of local group
    g - ahead.  Whatever the interleaving of the workgroups' takes, every block in [0, nblocks) is handed out exactly once, a
    wave stops at its first block >= nblocks, and the counter stays below the bound the launcher reports."""
    rng = np.random.default_rng(5)
    sizes = [1, 2, 7, 8, 9, 15, 16, 17, 63, 64, 65, 2047, 2048, 2049, 4369, 8191, 17477, 69871]
    sizes += [int(v) for v in rng.integers(1, 40_000, size=12)]
    for nblocks in sizes:
        for wgs_max in (1, 2, 3, 8, 256, 304):
            s = fir.debug_fft_schedule(nblocks, wgs_max)
            assert s["RA"] == 8 and s["nA"] == (nblocks + 7) // 8 and s["RB"] in (1, 2)
            wgs, ahead = s["wgs"], s["RB"]
            assert 1 <= wgs <= wgs_max and wgs <= s["nA"]
            counter = 0
            seen = np.zeros(nblocks, dtype=np.int32)
            slots = [0] * wgs                       # LDS slot counter per workgroup
            ring = [{k: k * wgs + b for k in range(ahead)} for b in range(wgs)]
            live = [8] * wgs                        # waves still running per workgroup
            order = list(range(wgs))
            while any(live):
                b = int(rng.choice([w for w in order if live[w]]))
                sl = slots[b]
                slots[b] += 1
                g, j = divmod(sl, 8)
                if j == 0:
                    ring[b][g + ahead] = ahead * wgs + counter
                    counter += 1
                blk = ring[b][g] * 8 + j
                if blk >= nblocks:
                    live[b] -= 1                    # this wave is done
                else:
                    seen[blk] += 1
            assert (seen == 1).all(), (nblocks, wgs_max)
            assert counter <= s["tickets"], (nblocks, wgs_max, counter, s)


def test_block_queue_kernel_code_under_host_simulation():
    """tests/c/fft_queue_sim.cpp compiles qo-100-tools_amd/csrc/if_fir_fft_queue.h -- the queue code the kernel runs -- for
    the host and runs the waves of a launch as threads with random delays (also with the ticket fetches held back, the
    interleaving ADVICE r2 describes): every block handed out exactly once, every wave leaves, no bounded wait expires,
    the ticket counter stays below the launcher's bound."""
    import subprocess
    import tempfile
    exe = os.path.join(tempfile.mkdtemp(prefix="fft_queue_sim_"), "fft_queue_sim")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-I" + os.path.join(ROOT, "qo-100-tools_amd", "csrc"),
                           os.path.join(ROOT, "tests", "c", "fft_queue_sim.cpp"), "-o", exe])
    # (blocks, workgroups, hold the ticket fetches back); several have a tail (a remainder of at most one block per SIMD that
    # is kept out of the groups and handed out one block per SIMD, in launches of at most 6 two-wave rounds): 70/4, 100/3, 130/4, 200/4
    cases = [(1, 4, 0), (3, 4, 0), (8, 4, 0), (9, 4, 0), (17, 1, 0), (64, 8, 0), (65, 8, 1), (70, 4, 1), (100, 3, 1), (130, 4, 1),
             (33, 4, 1), (40, 4, 1), (17, 2, 1), (2185, 256, 1),      # one round of groups + a tail
             (200, 4, 0), (650, 4, 0), (1000, 6, 0), (1000, 6, 1), (2049, 8, 0), (4369, 5, 1), (8191, 8, 1), (69906, 8, 0)]
    for seed, (nblocks, wgs, delay) in enumerate(cases):
        run = subprocess.run([exe, str(nblocks), str(wgs), str(seed + 1), str(delay)], capture_output=True, text=True, timeout=300)
        assert run.returncode == 0 and run.stdout.strip().endswith("OK"), run.stdout + run.stderr
        if (nblocks, wgs) in ((70, 4), (100, 3), (130, 4), (200, 4), (33, 4), (40, 4), (17, 2), (2185, 256)):
            assert "+ tail 0)" not in run.stdout, run.stdout
        if (nblocks, wgs) in ((650, 4), (4369, 5), (69906, 8)):     # long launches: no tail phase
            assert "+ tail 0)" in run.stdout, run.stdout


def test_no_overlap_save_instantiation_spills():
    """The build records the compiler's per-kernel resource remarks (csrc/if_fir_fft.resources.txt).  No instantiation of
    the overlap-save kernel may use scratch: a spill reload behind row loads in flight waits for all of them (vmcnt is
    in order), which defeats the prefetch the kernel is built on.  Two waves per SIMD need <= 256 VGPRs."""
    path = os.path.join(ROOT, "qo-100-tools_amd", "csrc", "if_fir_fft.resources.txt")
    assert os.path.exists(path), "build() first: the Makefile writes this file next to if_fir_fft.o"
    kernels, cur = {}, None
    for line in open(path):
        k, _, v = line.strip().partition(":")
        if k == "Function Name":
            cur = v.strip()
            kernels[cur] = {}
        elif cur:
            kernels[cur][k.split("[")[0].strip()] = int(v)
    fft = {k: v for k, v in kernels.items() if "fir_fft_kernel" in k}
    # 5 overlap lengths x (4 + 4 + 4 + 4 single-channel: full rate, decimate-by-4, -by-2, selecting store; 2 + 4 + 4 filter-bank: decimation 4,
    # and 8 / 16 with and without NCO) variants + 8 accumulating ones
    # + 20 + 20 (round 3): the decimate-by-4 / -by-2 tails keeping every sub-th output (decimation 8, 12, ..., 64 / 6, 10, ..., 62)
    # + 16: the second partition of 3074..4096-tap filters behind the four single-channel tails (accumulating store, 32 rows)
    # (round 4: the bank at decimation 8 keeps two forms per input format: channels on the slot grid, and channels at any centre bin / an NCO)
    # + 10 (round 4): the bank at decimation 8, all slots of one parity (5 overlap lengths x float32 / int16)
    # + 10 (round 4): the bank at decimation 16 with every channel at its own centre (per channel, four per small inverse)
    # + 10 (round 4): the same at decimation 4 (per channel, one 1024-point inverse each)
    # + 10 (round 4): the decimation-4 general form keeping every sub-th output (decimation 12, 20, ...) as its own instantiation
    # + 10 (round 4): the all-slots form with the context's NCO (a common offset of the slot grid)
    assert len(fft) == 244, len(fft)
    # (round 5) 48 of them are the decimate-by-2 tails (CHAN 2, 3: 5 x 8 + 8 accumulating), instantiated in units of their own
    assert sum(1 for k in fft if re.search(r"Li[23]ELb0ELb[01]E", k)) == 48
    for name, res in fft.items():
        assert res["ScratchSize"] == 0 and res["VGPRs Spill"] == 0 and res["VGPRs"] <= 256, (name, res)
    # round 4: the odd-decimation kernel (blocks of 3 x 1024 samples): 2 overlap lengths x float32 / int16 x NCO x thinning
    odd = {k: v for k, v in kernels.items() if "fir_odd_kernel" in k}
    assert len(odd) == 16, len(odd)
    for name, res in odd.items():
        assert res["ScratchSize"] == 0 and res["VGPRs Spill"] == 0 and res["VGPRs"] <= 256, (name, res)


def _hazard_tool():
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_store_hazard", os.path.join(ROOT, "tools", "check_store_hazard.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_wide_store_hazard_scanner_flags_the_round_3_form_and_passes_the_build():
    """a store of more than 8 bytes followed too closely by a VALU write of its data registers corrupted
    outputs now and then in round 3 (16-byte store with the row offset in an SGPR: the compiler places no wait state).  The
    disassembly of every device object of the library is scanned (tools/check_store_hazard.py, also a build step); the
    scanner must flag the old form, compiled here from the kernel source with IF_FIR_FFT_HAZARD_PROBE=1, and pass the
    current one (=2: the same single instantiation) and everything the build produced."""
    import subprocess
    import tempfile
    tool = _hazard_tool()
    # synthetic snippets: 0 and 1 wait states are violations, 2 are enough; an s_nop 1 counts two; a non-overlapping write is fine
    st = "\tbuffer_store_dwordx4 v[10:13], v40, s[36:39], s5 offen nt\n"
    assert tool.scan_text(st + "\tv_pk_add_f32 v[12:13], v[2:3], v[4:5]\n")
    assert tool.scan_text(st + "\tv_add_u32_e32 v40, 0x400, v40\n\tv_mov_b32_e32 v10, 0\n")
    assert not tool.scan_text(st + "\tv_add_u32_e32 v40, 0x400, v40\n\ts_nop 0\n\tv_mov_b32_e32 v10, 0\n")
    assert not tool.scan_text(st + "\ts_nop 1\n\tv_pk_mul_f32 v[10:11], v[2:3], v[4:5]\n")
    assert not tool.scan_text(st + "\tv_pk_add_f32 v[14:15], v[2:3], v[4:5]\n\tv_pk_add_f32 v[8:9], v[2:3], v[4:5]\n")
    assert tool.scan_text("\tglobal_store_dwordx4 v[2:3], v[10:13], off\n\tv_permlane32_swap_b32_e32 v1, v11\n")
    # round 5: the scan goes on behind a conditional branch -- on the fall-through path (the branch is one wait state) and
    # at its target (a loop's back edge: the store at the end of the loop body, the VALU write at its head)
    assert tool.scan_text(st + "\ts_cbranch_vccnz .LBB0_9\n\tv_mov_b32_e32 v11, 0\n.LBB0_9:\n\ts_endpgm\n")
    assert not tool.scan_text(st + "\ts_cbranch_vccnz .LBB0_9\n\ts_nop 0\n\tv_mov_b32_e32 v11, 0\n.LBB0_9:\n\ts_endpgm\n")
    loop = ".LBB0_1:\n\tv_pk_add_f32 v[12:13], v[2:3], v[4:5]\n\ts_add_u32 s4, s4, 1\n\ts_cmp_lt_u32 s4, s5\n" + st
    assert tool.scan_text(loop + "\ts_cbranch_scc1 .LBB0_1\n\ts_endpgm\n")               # back edge: 1 wait state, then the write
    assert not tool.scan_text(loop + "\ts_nop 0\n\ts_cbranch_scc1 .LBB0_1\n\ts_endpgm\n")
    assert tool.scan_text(st + "\ts_branch .LBB0_7\n\ts_nop 4\n.LBB0_7:\n\tv_mov_b32_e32 v13, 1.0\n")       # unconditional: the target
    assert tool.scan_text(st + "\ts_cbranch_execz .Lnowhere\n\ts_nop 4\n")                                    # unresolvable target
    # the same through the addresses of a disassembly
    dis = ("0000000000001000 <kern>:\n\tv_pk_add_f32 v[12:13], v[2:3], v[4:5]  // 000000001000: D3B2000C\n"
           "\tbuffer_store_dwordx4 v[10:13], v40, s[36:39], 0 offen  // 000000001008: E07C1000\n"
           "\ts_cbranch_scc1 65532  // 000000001010: BF85FFFC <kern+0x0>\n\ts_endpgm  // 000000001014: BF810000\n")
    assert tool.scan_text(dis)
    # a bundle without device code is an error, not "0 wide stores, 0 violations"
    import isa_tools
    with tempfile.TemporaryDirectory() as td:
        hostonly = os.path.join(td, "host.o")
        open(os.path.join(td, "h.c"), "w").write("int f(void) { return 1; }\n")
        subprocess.check_call(["gcc", "-c", os.path.join(td, "h.c"), "-o", hostonly])
        try:
            tool.disassemble(hostonly)
            raise AssertionError("a host-only object must not disassemble to an empty text")
        except (isa_tools.NoDeviceCode, subprocess.CalledProcessError):
            pass
        assert tool.main([hostonly]) == 1
    csrc = os.path.join(ROOT, "qo-100-tools_amd", "csrc")
    with tempfile.TemporaryDirectory() as td:
        for probe, expect_bad in ((1, True), (2, False)):
            out = os.path.join(td, "probe%d.s" % probe)
            subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                                   "-I" + csrc, "-DIF_FIR_FFT_ROWS=4", "-DIF_FIR_FFT_HAZARD_PROBE=%d" % probe, "-S", "--cuda-device-only",
                                   os.path.join(csrc, "if_fir_fft.hip"), "-o", out], stderr=subprocess.DEVNULL)
            text = open(out).read()
            assert tool.count_wide_stores(text) >= 12, "the probe no longer holds the 16-byte stores of the decimate-by-2 tail"
            bad = tool.scan_text(text, "probe%d" % probe)
            assert bool(bad) == expect_bad, (probe, bad[:3])
    # (round 5: the decimate-by-2 tails, the only kernels of the overlap-save source with 16-byte stores, live in units of their own)
    rows = (4, 8, 16, 32, 48)
    objs = ["if_fir_fft_r%d.o" % r for r in rows] + ["if_fir_fft_d2_r%d.o" % r for r in rows] + ["if_fir_fft_odd.o", "if_fir_kernels.o", "wb_detect.o"]
    for o in objs:
        path = os.path.join(csrc, o)
        assert os.path.exists(path), "build() first: %s" % o
        text = tool.disassemble(path)
        assert tool.count_wide_stores(text) > 0 or o == "if_fir_fft_odd.o" or o.startswith("if_fir_fft_r"), o   # (8 bytes per lane there)
        assert not tool.scan_text(text, o), o


def _exchange_tool():
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_lds_exchange", os.path.join(ROOT, "tools", "check_lds_exchange.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_lds_exchange_order_gate_flags_a_reordered_probe_and_passes_the_build():
    """the lane exchanges of the overlap-save kernels (16 stores per lane into the wave's LDS buffer, 16 loads, the next
    exchange's stores) are correct only while the instruction stream keeps the phases apart; for the compiler they are plain
    loads and stores of one thread, and for most pairs an alias-freedom proof exists.  The source now routes every exchange load
    through an opaque base address (no proof possible: the language pins the order), and the build walks every kernel's
    line-annotated disassembly (tools/check_lds_exchange.py).  Here: synthetic streams, a deliberately mis-ordered probe compiled from
    the kernel source (IF_FIR_FFT_LDSX_PROBE=1: the last store of a phase behind its first load) must be flagged, the correct
    one and every unit of the build must pass, and the line tables the classification needs must not change the code."""
    import subprocess
    import tempfile
    tool = _exchange_tool()
    csrc = os.path.join(ROOT, "qo-100-tools_amd", "csrc")
    src = os.path.join(csrc, "if_fir_fft.hip")
    tags = tool.tagged_lines(src)          # over if_fir_fft.hip and the parts it includes: {(file name, line): 'W' | 'R'}
    fw, lw = [k for k, v in tags.items() if v == "W"][0]
    fr, lr = [k for k, v in tags.items() if v == "R"][0]
    fw, fr = os.path.join(csrc, fw), os.path.join(csrc, fr)
    assert len(tags) >= 12     # the two helpers and every call site
    assert {k[0] for k in tags} >= {"if_fir_fft_dev.h", "if_fir_fft_odd.inc"}

    def stream(*parts):
        out = ["0000000000001000 <_ZN6if_fir14fir_fft_kernelILi4EEEvv>:"]
        for kind, n in parts:
            if kind == "W":
                out.append("; %s:%d" % (fw, lw))
                out += ["\tds_write_b64 v1, v[2:3] offset:%d  // 000000001000: 0" % (136 * k) for k in range(n)]
            elif kind == "R":
                out.append("; %s:%d" % (fr, lr))
                out += ["\tds_read2_b64 v[4:7], v9 offset0:%d offset1:%d  // 000000001000: 0" % (2 * k, 2 * k + 1) for k in range(n // 2)]
                out += ["\tds_read_b64 v[4:5], v9  // 000000001000: 0"] * (n % 2)
            elif kind == "T":    # table reads: another source line, ignored wherever they stand
                out.append("; %s:%d" % (src, 1))
                out += ["\tds_read2_b64 v[4:7], v8 offset0:4 offset1:8  // 000000001000: 0"] * n
            else:                # a DS instruction without a source line
                out.append("; %s:0" % src)
                out += ["\tds_read_b64 v[4:5], v9  // 000000001000: 0"]
        return "\n".join(out) + "\n"

    base = os.path.basename(src)
    ok = lambda *parts: tool.scan_text(stream(*parts), tags, base)[0]
    assert not ok(("W", 16), ("T", 3), ("R", 16), ("W", 16), ("R", 16))
    assert not ok(("W", 8), ("T", 2), ("W", 8), ("R", 6), ("T", 1), ("R", 10))
    assert ok(("W", 15), ("R", 2), ("W", 1), ("R", 14))                 # a load ahead of the phase's last store
    assert ok(("W", 16), ("R", 14), ("W", 16), ("R", 18))               # the next exchange's stores ahead of the last loads
    assert ok(("W", 16), ("W", 16), ("R", 16), ("R", 16))               # two exchanges interleaved
    assert ok(("W", 16), ("R", 16), ("W", 16))                          # ends inside an exchange
    assert ok(("T", 4))                                                 # a kernel of the family without any classified access
    assert ok(("W", 16), ("X", 1), ("R", 16))                           # a DS instruction without a source line
    flags = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-fvisibility=hidden", "-I" + os.path.join(ROOT, "include"),
             "-I" + csrc, "-DIF_FIR_FFT_ROWS=4", "-DIF_FIR_FFT_HAZARD_PROBE=2"]
    with tempfile.TemporaryDirectory() as td:
        for probe, expect_bad in ((1, True), (0, False)):
            out = os.path.join(td, "probe%d.o" % probe)
            subprocess.check_call(flags + ["-gline-tables-only", "-DIF_FIR_FFT_LDSX_PROBE=%d" % probe, "-c", src, "-o", out], stderr=subprocess.DEVNULL)
            bad, units = tool.scan_text(tool.isa_tools.disassemble(out, lines=True), tags, base, "probe%d" % probe)
            assert bool(bad) == expect_bad, (probe, bad[:3])
            assert sum(units.values()) >= 8, units
        # line tables do not change the generated code: the same unit without them disassembles to the same instructions
        plain = os.path.join(td, "plain.o")
        subprocess.check_call(flags + ["-DIF_FIR_FFT_LDSX_PROBE=0", "-c", src, "-o", plain], stderr=subprocess.DEVNULL)
        strip = lambda t: [ln.split("//")[0].rstrip() for ln in t.splitlines() if ln.startswith("\t")]
        a = strip(tool.isa_tools.disassemble(plain))
        b = strip(tool.isa_tools.disassemble(os.path.join(td, "probe0.o")))
        assert a == b and len(a) > 3000
    rows = (4, 8, 16, 32, 48)
    for o in ["if_fir_fft_r%d.o" % r for r in rows] + ["if_fir_fft_d2_r%d.o" % r for r in rows] + ["if_fir_fft_odd.o"]:
        path = os.path.join(csrc, o)
        assert os.path.exists(path), "build() first: %s" % o
        bad, units = tool.scan_text(tool.isa_tools.disassemble(path, lines=True), tags, base, o)
        fam = {k: n for k, n in units.items() if any(f in k for f in tool.FAMILY) and not k.endswith(".kd")}
        assert not bad, (o, bad[:3])
        assert len(fam) >= (8 if "_d2_" in o else 16) and min(fam.values()) >= 4, (o, len(fam))    # every kernel: exchange 2 (four rounds) at least
    # the fence macro of round 4 is gone: the order no longer depends on a switch
    assert not any("LDS_FENCE" in open(f).read() for f in tool.source_files(src))


def test_lds_reads_are_single_in_the_built_kernels():
    """Round 5 (profiles/r05_lds_single_reads.txt): a ds_read2_b64 / ds_read2st64_b64 pair takes 8 LDS cycles on 32 banks, two ds_read_b64 take 2 each on
    64 -- the overlap-save kernels are compiled so that the compiler forms no pairs (a per-kernel target attribute + the IR vectorizer off for their
    units).  Checked in the built objects: a toolchain that ignored the attribute would still be correct, only slower -- this test is what would notice."""
    import collections
    import sys
    csrc = os.path.join(ROOT, "qo-100-tools_amd", "csrc")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_tools

    def read_forms(obj):
        per, kern = collections.defaultdict(collections.Counter), None
        for ln in isa_tools.disassemble(os.path.join(csrc, obj)).splitlines():
            lab = isa_tools.label(ln)
            if lab is not None and not lab.startswith(".L") and "+0x" not in lab:
                kern = lab
                continue
            d = isa_tools.instr(ln)
            if d and kern and d[0].startswith("ds_read"):
                per[kern][d[0]] += 1
        return {k: v for k, v in per.items() if not k.endswith(".kd")}
    for obj, name, count in (("if_fir_fft_r4.o", "fir_fft_kernel", 36), ("if_fir_fft_r32.o", "fir_fft_kernel", 52), ("if_fir_fft_d2_r4.o", "fir_fft_kernel", 8),
                             ("if_fir_fft_d2_r32.o", "fir_fft_kernel", 16), ("if_fir_fft_odd.o", "fir_odd_kernel", 16)):
        assert os.path.exists(os.path.join(csrc, obj)), "build() first: %s" % obj
        forms = {k: v for k, v in read_forms(obj).items() if name in k}
        assert len(forms) == count, (obj, len(forms))
        for k, v in forms.items():
            assert set(v) == {"ds_read_b64"} and v["ds_read_b64"] >= 200, (obj, k[:90], dict(v))


def test_bench_line_contract_on_the_committed_run():
    """The bench.py JSON line of the last profiled run (profiles/*_bench.json, produced on the GPU box) carries every
    field the driver's contract names, with the right types, and its numbers are mutually consistent."""
    import glob
    import json
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench.json")))
    assert paths, "no committed bench line under profiles/"
    line = [ln for ln in open(paths[-1]).read().splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                     ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(d[key], typ), (key, type(d[key]))
    assert d["vs_baseline"] is None and d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic"
    assert d["dtype"] == "f32" and d["unit"] == "MSamples/s" and "255-tap" in d["metric"]
    assert "workload" in d["config"] and "2^28" in d["config"]["workload"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) < 1.0
    assert r["traffic"] is None or 0.9 * r["algorithmic_bytes_per_launch"] < r["traffic"] < 1.3 * r["algorithmic_bytes_per_launch"]
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and isinstance(c["sample"], str)
    # round 3: the cold figure (same steps before any conditioning) is printed beside the settled one
    assert d["cold_ms_per_step"] >= 0.9 * d["ms_per_step"] and 0.3 < r["cold_frac"] <= 1.1 * r["frac"]
    assert abs(r["cold_frac"] - r["algorithmic_bytes_per_launch"] / (r["cold_kernel_ms"] * 1e-3) / 1e9 / r["peak"]) < 1e-3
    for cfg in ("fir127_2p26", "fir1023_2p28"):
        auto = d["extra"]["configs"][cfg]["auto"]
        if "traffic" in auto:   # (present once profiles/traffic.json holds the config's counter passes)
            assert 0.95 < auto["traffic_over_algorithmic"] < 1.2 and auto["traffic"] > 0
    # value = samples of all ranks / wall time per step
    n = d["config"]["samples_per_channel"] * d["config"].get("channels", d["n_gpus"])
    assert abs(d["value"] - n / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 1e-3
    assert d["parity"]["whole_output"]["ok"] is True


def test_keep_every_index_arithmetic_of_the_decimating_tail():
    """Decimation 8, 12, ..., 64 run behind the decimate-by-4 tail, which keeps every sub-th of its outputs
    (csrc/if_fir_fft.hip, KeepEvery).  Its integer arithmetic, restated here, against plain division over the whole range
    the kernel can see: the block's share blk * lout divided in 32-bit pieces (blk < 2^31), and the per-output
    multiply-shift by ceil(2^18 / sub) for every numerator remainder + lane + 64 * row that can occur."""
    rng = np.random.default_rng(5)
    for sub in range(2, 32):    # up to 16 behind the decimate-by-4 tail, odd ones up to 31 behind the decimate-by-2 tail
        magic = (262144 + sub - 1) // sub
        u = np.arange(0, 32 + 128 + 128 * 15 + 1, dtype=np.uint64)  # remainder + lane offset (2 per lane + 1) + row step
        assert np.array_equal((u * np.uint64(magic)) >> np.uint64(18), u // np.uint64(sub)), sub
        assert int(u.max()) * magic < 2 ** 32        # the product stays in 32 bits
        for lout in (960, 896, 768, 512, 256, 1920, 1792, 1536, 1024):   # tail outputs per block: (4096 - 64 rows) / 4 or / 2
            blks = np.concatenate([np.arange(0, 70), rng.integers(0, 2 ** 31, 2000), [2 ** 31 - 1]]).astype(np.uint64)
            for blk in blks:
                blk = int(blk)
                if sub & (sub - 1) == 0:
                    q, rem = (blk * lout) >> (sub.bit_length() - 1), (blk * lout) & (sub - 1)
                else:
                    bq, br = (blk & 0xffffffff) // sub, (blk & 0xffffffff) % sub
                    t = (br * lout) & 0xffffffff
                    q2 = t // sub
                    q, rem = bq * lout + q2, t - q2 * sub
                assert (q, rem) == divmod(blk * lout, sub), (sub, lout, blk)
        # round 5 (KeepEvery::store_offset): both factors of both products fit 24 bits (v_mul_u32_u24, full rate), and the byte offset from the
        # block's first kept output, (index - qb) * 8 with qb = q + (rem ? 1 : 0), is (qd - (rem ? 1 : 0)) * 8 in 32-bit arithmetic -- never negative
        assert int(u.max()) < 2 ** 24 and magic < 2 ** 24 and sub < 2 ** 24
        for rem in range(sub):
            qd = (u + np.uint64(rem)) // np.uint64(sub)
            kept = (u + np.uint64(rem)) % np.uint64(sub) == 0
            q = 12345678901                                      # any block quotient: it cancels
            index = q + qd[kept].astype(np.int64)
            qb = q + (1 if rem else 0)
            off32 = (qd[kept].astype(np.int64) - (1 if rem else 0)) * 8
            assert np.array_equal((index - qb) * 8, off32) and (off32 >= 0).all() and int(off32.max()) < 2 ** 32, (sub, rem)


def test_filter_bank_decimation_8_routing(fir):
    """launch_fft_rows' routing of a decimation-8 bank call on the slot grid (fft_bank8_plan, host-only): a slot parity with at least
    four channels, no slot of the call listed twice, goes through ONE all-slots launch; everything else per channel."""
    even, odd = 0x5555, 0xAAAA
    assert fir.debug_bank8_plan(list(range(16))) == (even, odd, [])
    assert fir.debug_bank8_plan([5, 0, 15, 8, 3, 10, 3]) == (0, 0, list(range(7)))          # a slot twice: per channel
    assert fir.debug_bank8_plan([1, 5, 9, 13, 2]) == (0, (1 << 1) | (1 << 5) | (1 << 9) | (1 << 13), [4])
    assert fir.debug_bank8_plan([7]) == (0, 0, [0])
    assert fir.debug_bank8_plan([0, 2, 4]) == (0, 0, [0, 1, 2])                              # fewer than four of a parity
    assert fir.debug_bank8_plan([0, 2, 4, 14, 1, 3, 5]) == (0x4015, 0, [4, 5, 6])
    assert fir.debug_bank8_plan([3, 1, 15, 13, 11, 9, 7, 5, 0, 4]) == (0, odd, [8, 9])
    rng = np.random.default_rng(8)
    for _ in range(200):
        k = int(rng.integers(1, 17))
        slots = [int(v) for v in (rng.permutation(16)[:k] if rng.random() < 0.6 else rng.integers(0, 16, size=k))]
        me, mo, rest = fir.debug_bank8_plan(slots)
        assert me & odd == 0 and mo & even == 0
        served = [c for c in range(k) if ((me | mo) >> slots[c]) & 1]
        assert sorted(served + rest) == list(range(k))                                      # every channel exactly once
        dup = len(set(slots)) != k
        for par, m in ((0, me), (1, mo)):
            want = [s for s in slots if s % 2 == par]
            assert m == (sum(1 << s for s in set(want)) if (not dup and len(want) >= 4) else 0), (slots, par, m)


def test_filter_bank_tail_by_decimation(fir):
    """fft_bank_tail (host-only): the slot API serves decimation 4, 8, 16; channels at their own centres every multiple of 4 up to 64,
    behind the tail of the largest of 16, 8, 4 that divides the decimation."""
    for d in range(0, 70):
        slot = d if d in (4, 8, 16) else 0
        own = 0 if (d < 4 or d > 64 or d % 4) else 16 if d % 16 == 0 else 8 if d % 8 == 0 else 4
        assert fir.debug_bank_tail(d) == slot, d
        assert fir.debug_bank_tail(d, True) == own, d


def test_filter_bank_table_images(fir):
    """The decimation-8 bank's two table images (fft_build_tables, host-only) against their definition in float64:
    G_q[a] = W16^(a q) sum_j H(k_low + 256 (q + 2 j)) W8^(a j), times b^a (b = W4096^k_low: the factor input a of the last forward
    pass still carries); the all-slots form's image for the ODD slots = the same times W16^a with the halves exchanged.  And the
    16-slot image: G0[n2] = sum_k2 H(k_low + 256 k2) W16^(n2 k2), times b^n2."""
    rng = np.random.default_rng(12)
    h = rng.standard_normal(255).astype(np.float32)
    H = np.fft.fft(h.astype(np.float64), 4096) / 4096.0
    lane = np.arange(64)
    W = lambda n, e: np.exp(-2j * np.pi * (np.asarray(e) % n) / n)   # noqa: E731
    ev = fir.debug_fft_tables_bank(h, 8, 0)["hp"].astype(np.complex128)
    od = fir.debug_fft_tables_bank(h, 8, 1)["hp"].astype(np.complex128)
    g16 = fir.debug_fft_tables_bank(h, 16)["hp"].astype(np.complex128)
    scale = np.abs(H).max() * 8
    for i in range(4):
        klow = (4 * (lane // 16) + i) + 16 * (lane % 16)
        for q in range(2):
            for a in range(8):
                g = W(16, a * q) * sum(H[klow + 256 * (q + 2 * j)] * W(8, a * j) for j in range(8)) * W(4096, a * klow)
                assert np.max(np.abs(ev[(i * 16 + 8 * q + a) * 64 + lane] - g)) <= 2e-7 * scale, (i, q, a)
                assert np.max(np.abs(od[(i * 16 + 8 * (1 - q) + a) * 64 + lane] - g * W(16, a))) <= 2e-7 * scale, (i, q, a)
        for n2 in range(16):
            g = sum(H[klow + 256 * k2] * W(16, n2 * k2) for k2 in range(16)) * W(4096, n2 * klow)
            assert np.max(np.abs(g16[(i * 16 + n2) * 64 + lane] - g)) <= 2e-7 * scale * 2, (i, n2)


def test_phasor_tables_and_the_banks_small_inverse_table(fir):
    """Round 5: every (cos, tan) table image carries the phasor tables P1[k] = exp(j 2 pi k / 2^7), P2[k] = exp(j 2 pi k / 2^14) at bytes
    [6 KB, 8 KB) (lds_phasor: a 32-bit phase -> phasor with two table reads and a second-order polynomial for the low 18 bits,
    instead of two sincospif); the numpy model of that arithmetic stays within 2e-7 of the exact phasor over random and edge
    phases.  The filter-bank images hold the twiddles between the two transforms of their small inverse as (cos, tan) entries of
    b = W256^mu1 (inverse_tail256_tan)."""
    taps = fir.bpf_design(255, 0.0, 0.02)
    imgs = [fir.debug_fft_tables(taps, 4), fir.debug_fft_tables(taps, 1), fir.debug_fft_tables(taps, 8),
            fir.debug_fft_tables_bank(taps, 8), fir.debug_fft_tables_bank(taps, 16), fir.debug_fft_tables_bank(taps, 8, parity=1)]
    k = np.arange(128)
    p1 = np.exp(2j * np.pi * k / 128.0).astype(np.complex64)
    p2 = np.exp(2j * np.pi * k / 16384.0).astype(np.complex64)
    for im in imgs:
        pht = im["tw1"][768:1024]          # bytes 6144 .. 8191 of the image
        assert np.array_equal(pht[:128], p1) and np.array_equal(pht[128:], p2)
    odd = fir.debug_fft_tables_odd(taps, 3)["pht"]     # the odd-decimation kernel's image: behind its NCO row phasors
    assert np.array_equal(odd[:128], p1) and np.array_equal(odd[128:], p2)
    # the kernel's arithmetic in float32 (lds_phasor)
    rng = np.random.default_rng(5)
    ph = np.concatenate([rng.integers(0, 2 ** 32, 200000, dtype=np.uint64), np.array([0, 1, 2 ** 18 - 1, 2 ** 18, 2 ** 25 - 1, 2 ** 32 - 1], dtype=np.uint64)])
    a = p1[(ph >> 25).astype(np.int64)]
    b = p2[((ph >> 18) & 127).astype(np.int64)]
    th = ((ph & 0x3FFFF).astype(np.float32) * np.float32(1.4629180792671596e-9)).astype(np.float32)
    lo = (np.float32(1.0) - np.float32(0.5) * th * th).astype(np.float32) + 1j * th
    got = ((a * b).astype(np.complex64) * lo.astype(np.complex64)).astype(np.complex64)
    exact = np.exp(2j * np.pi * ph.astype(np.float64) / 2.0 ** 32)
    assert np.max(np.abs(got - exact)) < 2e-7
    # the banks' table between the two transforms of the small inverse: entry e of b = W256^mu1 at [e * 16 + mu1]
    for im in imgs[3:]:
        twe = im["twe"]
        for mu1 in (0, 1, 5, 15):
            th0 = -2.0 * np.pi * mu1 / 256.0
            def ent(theta, cref=0.0):
                c = np.cos(theta)
                if abs(c) < 9.3e-10:
                    c = np.copysign(9.313225746154785e-10, c)
                c = float(np.float32(c))
                return np.float32(c / cref if cref else c), np.float32(np.sin(theta) / c), c
            e0 = ent(4 * th0)
            assert twe[0 * 16 + mu1].real == e0[0] and twe[0 * 16 + mu1].imag == e0[1]
            e2 = ent(12 * th0, e0[2])
            assert twe[2 * 16 + mu1].real == e2[0] and twe[2 * 16 + mu1].imag == e2[1]
            for q in range(4):
                bq = th0 - 2.0 * np.pi * q / 16.0
                e = ent(bq)
                assert twe[(3 + 3 * q) * 16 + mu1].real == e[0] and twe[(3 + 3 * q) * 16 + mu1].imag == e[1]
        assert not np.any(twe[15 * 16:])   # the rest of the 8 KB slot is unused


def test_host_table_builders_under_sanitizers(tmp_path):
    """The HOST side of csrc/if_fir_fft.hip -- fft_build_tables for every image (single-channel, full-rate, bank 8 even / odd, bank
    16), fft_build_tables_odd, the decimation-8 routing -- compiled with AddressSanitizer + UBSan (CPU only: hipcc --offload-host-only)
    and run over taps 1 ... 4096, real and complex, into heap buffers of exactly the documented sizes (tests/c/host_tables_asan.cpp)."""
    import subprocess
    src = os.path.join(ROOT, "qo-100-tools_amd", "csrc", "if_fir_fft.hip")
    inc = ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "qo-100-tools_amd", "csrc")]
    san = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-std=c++17"]
    obj = str(tmp_path / "fft_host.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-host-only"] + san + inc + ["-c", src, "-o", obj])
    exe = str(tmp_path / "host_asan")
    subprocess.check_call(["/opt/rocm/lib/llvm/bin/clang++"] + san + inc + ["-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__",
                           os.path.join(ROOT, "tests", "c", "host_tables_asan.cpp"), obj, "-o", exe,
                           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"])
    run = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert run.returncode == 0 and "host table builders: clean" in run.stdout, (run.stdout[-2000:], run.stderr[-4000:])
