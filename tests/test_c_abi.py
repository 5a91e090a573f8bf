"""The C ABI from C: tests/c/if_fir_selftest.c is compiled with gcc against include/if_fir.h and libif_fir.so (no HIP
headers, no Python in the loop) and run on the GPU."""
import os
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_c_host_program_on_the_c_abi(gpu_ok):
    libdir = os.path.join(ROOT, "qo-100-tools_amd")
    assert os.path.exists(os.path.join(libdir, "libif_fir.so")), "libif_fir.so is not built"
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "if_fir_selftest")
        subprocess.check_call(["gcc", "-std=c99", "-O1", "-Wall", "-Wextra", "-I" + os.path.join(ROOT, "include"),
                               os.path.join(ROOT, "tests", "c", "if_fir_selftest.c"), "-L" + libdir, "-lif_fir", "-lm",
                               "-Wl,-rpath," + libdir, "-o", exe])
        run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
        assert run.returncode == 0, run.stdout + run.stderr
        assert "all checks passed" in run.stdout


@pytest.mark.gpu
def test_stream_filter_program(gpu_ok):
    """qo-100-tools_amd/host/if_fir_pipe: stdin -> stdout stream filter (C, on the C ABI).  int16 samples in, NCO mix,
    255 taps, decimate by 4, fed in calls of 4096 samples; the bytes coming out against the float64 oracle."""
    import numpy as np
    import __graft_entry__ as g
    oracle = g.load_oracle()
    fir = g.load_pkg().if_fir
    exe = os.path.join(ROOT, "qo-100-tools_amd", "host", "if_fir_pipe")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.dirname(exe), "if_fir_pipe"])
    n = 100_003
    xi = np.clip(np.round(oracle.synth_iq(n, 7) * 14000.0), -32768, 32767).astype(np.int16)
    run = subprocess.run([exe, "-t", "255", "-d", "4", "-b", "0.0:0.05", "-n", "0.2", "-i", "s16", "-c", "4096"],
                         input=xi.tobytes(), capture_output=True, timeout=300)
    assert run.returncode == 0, run.stderr.decode()
    y = np.frombuffer(run.stdout, dtype=np.float32)
    taps = fir.bpf_design(255, 0.0, 0.05)
    ref = oracle.fir_nco_f64(taps, xi.astype(np.float32) * np.float32(2.0 ** -15), 4, oracle.nco_phase_word(0.2))
    assert y.shape == ref.shape
    l2, mx = oracle.err_metrics(y, ref)
    assert l2 <= 1e-6 and mx <= 1e-6, (l2, mx)
    assert b"100003 samples in, 25001 out" in run.stderr


@pytest.mark.gpu
def test_channelizer_program(gpu_ok, tmp_path):
    """qo-100-tools_amd/host/if_fir_channelize: stdin -> one file per channel (C, on the C ABI): three narrow channels at their own
    centres out of one int16 stream, decimated by 64, fed in calls of 8192 samples; every file against the float64 NCO oracle."""
    import numpy as np
    import __graft_entry__ as g
    oracle = g.load_oracle()
    fir = g.load_pkg().if_fir
    exe = os.path.join(ROOT, "qo-100-tools_amd", "host", "if_fir_channelize")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.dirname(exe), "if_fir_channelize"])
    n, d = 200_003, 64
    centres = [440 / 4096.0, -823 / 4096.0, 0.25]
    xi = np.clip(np.round(oracle.synth_iq(n, 9) * 14000.0), -32768, 32767).astype(np.int16)
    run = subprocess.run([exe, "-t", "255", "-d", str(d), "-w", "0.006", "-f", ",".join("%.12f" % c for c in centres),
                          "-o", str(tmp_path / "nb_%u.cf32"), "-i", "s16", "-c", "8192"],
                         input=xi.tobytes(), capture_output=True, timeout=300)
    assert run.returncode == 0, run.stderr.decode()
    taps = fir.bpf_design(255, 0.0, 0.006)
    x = xi.astype(np.float32) * np.float32(2.0 ** -15)
    for c, fc in enumerate(centres):
        y = np.fromfile(str(tmp_path / ("nb_%u.cf32" % c)), dtype=np.float32)
        ref = oracle.fir_nco_f64(taps, x, d, oracle.nco_phase_word(fc))
        assert y.shape == ref.shape, (c, y.shape, ref.shape)
        l2, mx = oracle.err_metrics(y, ref)
        assert l2 <= 1e-6 and mx <= 1e-6, (c, l2, mx)
    assert b"200003 samples in, 3126 out per channel, 3 channels" in run.stderr


def test_channelizer_program_validates_its_arguments_before_any_arithmetic():
    """`-d 0` or a non-numeric `-d` divided by zero, negative counts wrapped to billions, and the `-o` pattern went to
    snprintf as it came.  The program now refuses such arguments with exit code 2 before it touches the library (CPU test: no
    GPU is needed to be told that the arguments are bad)."""
    exe = os.path.join(ROOT, "qo-100-tools_amd", "host", "if_fir_channelize")
    assert os.path.exists(exe), "build() first"
    for bad in (["-d", "0"], ["-d", "x"], ["-d", "6"], ["-d", "128"], ["-t", "-5"], ["-t", "0"], ["-c", "5"], ["-c", "-1"],
                ["-o", "a%s"], ["-o", "nothing"], ["-o", "a%u%u"], ["-o", "a%n%u"], ["-w", "0"], ["-w", "0.7"]):
        run = subprocess.run([exe, "-f", "0.1"] + bad, stdin=subprocess.DEVNULL, capture_output=True, text=True, timeout=60)
        assert run.returncode == 2 and "bad argument" in run.stderr, (bad, run.returncode, run.stderr[:200])
