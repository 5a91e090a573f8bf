"""WB-transponder signal detector: the one routine of the reference that runs, so the one place where
parity is PINNED — tests/golden/wb_detect_golden.json was captured by running the reference's detect_signals() under
node (tests/golden/make_wb_golden.js).  CPU: the oracle restatement against those vectors, bit for bit.  GPU: the
batched HIP kernel against the vectors and against the oracle on many random frames, bit for bit."""
import base64
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def wb_gold():
    return json.load(open(os.path.join(HERE, "golden", "wb_detect_golden.json")))


@pytest.fixture(scope="module")
def wbo():
    from oracle import wb_oracle
    return wb_oracle


def _num(v):
    return float(v) if isinstance(v, str) else v        # "NaN"/"Infinity" were written as strings


def _same(a, b):
    a, b = float(a), float(_num(b))
    return (np.isnan(a) and np.isnan(b)) or a == b


def check_frame(wbo, frame, signals, expected):
    """frame record + signals array (structured numpy) against the reference's JSON result, exact float64 equality."""
    assert _same(frame["noise_power"], expected["noise_power"])
    assert bool(frame["beacon_valid"]) == ("beacon" in expected)
    assert int(frame["signal_count"]) == len(expected["signals"])
    pairs = [(signals[k], e) for k, e in enumerate(expected["signals"])]
    if "beacon" in expected:
        pairs.append((frame["beacon"], expected["beacon"]))
    for got, exp in pairs:
        for k in wbo.SIGNAL_FIELDS:
            assert _same(got[k], exp[k]), (k, float(got[k]), exp[k])
        assert bool(got["out_of_band"]) == exp["out_of_band"] and bool(got["over_powered"]) == exp["over_powered"]


def test_oracle_matches_the_reference_vectors(wbo, wb_gold):
    assert len(wb_gold["frames"]) >= 60 and "wb_spectrum_monitor.js" in wb_gold["source"]
    seen_beacon = seen_many = seen_oob = seen_over = 0
    for rec in wb_gold["frames"]:
        bins = np.frombuffer(base64.b64decode(rec["bins"]), dtype="<u2")
        assert "error" not in rec
        st, frame, sig = wbo.detect(bins)
        assert st == 1
        check_frame(wbo, frame, sig, rec["result"])
        seen_beacon += "beacon" in rec["result"]
        seen_many += len(rec["result"]["signals"]) >= 4
        seen_oob += any(s["out_of_band"] for s in rec["result"]["signals"])
        seen_over += any(s["over_powered"] for s in rec["result"]["signals"])
    assert seen_beacon and seen_many and seen_oob and seen_over     # the fixture exercises every branch
    assert wbo.detect(np.zeros(0, dtype="<u2"))[0] == 0             # the reference rejects an empty frame
    # capacity: the count is the true number of signals, only the stored part is bounded
    noisy = np.frombuffer(base64.b64decode(wb_gold["frames"][6]["bins"]), dtype="<u2")
    st, frame, sig = wbo.detect(noisy, max_signals=5)
    assert st == 1 and frame["signal_count"] == len(wb_gold["frames"][6]["result"]["signals"]) and sig.size == 5


def _random_frames(rng, n_frames, n_bins):
    """Frames in the style of the fixture: noise floor, optional beacon, a few trapezoid signals."""
    f = rng.uniform(6000, 14000, size=(n_frames, n_bins))
    for k in range(n_frames):
        if rng.random() < 0.7 and n_bins > 300:
            a = int(rng.integers(40, 100))
            f[k, a:a + 150] = np.maximum(f[k, a:a + 150], rng.uniform(30000, 55000) + rng.uniform(-500, 500, 150))
        for _ in range(int(rng.integers(0, 7))):
            a = int(rng.integers(0, n_bins))
            w = int(2 + rng.random() ** 2 * min(260, n_bins))
            p = rng.uniform(16000, 60000)
            seg = f[k, a:a + w]
            ramp = np.minimum(1.0, (np.minimum(np.arange(seg.size), seg.size - 1 - np.arange(seg.size)) + 1) / max(1, int(rng.integers(1, 12))))
            f[k, a:a + w] = np.maximum(seg, 12000 + (p - 12000) * ramp + rng.uniform(-1500, 1500, seg.size))
    return np.clip(np.round(f), 0, 65535).astype("<u2")


@pytest.mark.gpu
def test_gpu_detector_matches_the_reference_vectors(pkg, wbo, wb_gold, gpu_ok):
    """The HIP kernel through the C-ABI against the vectors captured from the reference, bit for bit (frames of the
    fixture have several lengths: one call per length)."""
    wb = pkg.wb_detect
    by_len = {}
    for rec in wb_gold["frames"]:
        bins = np.frombuffer(base64.b64decode(rec["bins"]), dtype="<u2")
        by_len.setdefault(bins.size, []).append((bins, rec["result"]))
    assert 918 in by_len and len(by_len) >= 8
    for n_bins, items in by_len.items():
        frames, signals = wb.detect_frames(np.stack([b for b, _ in items]), max_signals=128)
        for k, (_, expected) in enumerate(items):
            check_frame(wbo, frames[k], signals[k], expected)


@pytest.mark.gpu
@pytest.mark.parametrize("n_frames,n_bins", [(1, 918), (63, 918), (64, 918), (65, 918), (4099, 918), (300, 2047),
                                             (130, 12000), (9, 90001), (70, 5), (70, 2), (70, 1)])
def test_gpu_detector_matches_the_oracle_on_random_frames(pkg, wbo, gpu_ok, n_frames, n_bins):
    """Bit-for-bit against the pinned oracle: partial workgroups, LDS-staged frames of several lengths (odd row strides),
    frames too long for LDS (global-memory walk), degenerate lengths; signal capacity smaller than the signal count."""
    wb = pkg.wb_detect
    rng = np.random.default_rng(n_frames * 131 + n_bins)
    bins = _random_frames(rng, n_frames, n_bins)
    cap = 6
    frames, signals = wb.detect_frames(bins, max_signals=cap)
    assert frames.dtype.itemsize == 128 and signals.dtype.itemsize == 112
    checked = 0
    for k in (range(n_frames) if n_frames <= 130 else rng.choice(n_frames, 130, replace=False)):
        st, oframe, osig = wbo.detect(bins[k], max_signals=cap)
        assert st == 1
        assert frames[k].tobytes() == oframe.tobytes(), k
        stored = min(int(oframe["signal_count"]), cap)
        assert signals[k][:stored].tobytes() == osig[:stored].tobytes(), k
        checked += stored
    assert checked > 0 or n_bins < 20 or n_frames < 60       # enough frames: some of them must hold signals


@pytest.mark.gpu
def test_gpu_detector_argument_errors(pkg, gpu_ok):
    wb = pkg.wb_detect
    with pytest.raises(wb.WbDetectError, match="empty frame"):
        wb.detect_frames(np.zeros((3, 0), dtype="<u2"))
    frames, signals = wb.detect_frames(np.zeros((0, 918), dtype="<u2"))       # no frames: nothing to do
    assert frames.size == 0


@pytest.mark.gpu
def test_gpu_detector_every_noise_sample_value(pkg, wbo, gpu_ok):
    """The kernel forms the 3-bin average without a float64 division (q + fl(r/3), exact except for one sum that takes
    the division).  A 3-bin frame has exactly one averaged sample, so noise_power exposes it: every sum below the
    signal threshold, bit for bit against the oracle (which divides like the reference)."""
    wb = pkg.wb_detect
    sums = np.arange(0, 3 * 16500, dtype=np.int64)
    bins = np.stack([sums // 3, (sums + 1) // 3, (sums + 2) // 3], axis=1).astype("<u2")
    assert np.array_equal(bins.astype(np.int64).sum(axis=1), sums)
    frames, _ = wb.detect_frames(bins, max_signals=1)
    expect = np.array([wbo.detect(b, max_signals=1)[1]["noise_power"] for b in bins])
    assert np.array_equal(frames["noise_power"], expect)
    assert not frames["beacon_valid"].any() and not frames["signal_count"].any()
