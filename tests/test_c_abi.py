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
