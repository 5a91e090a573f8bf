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
