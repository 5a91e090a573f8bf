"""GPU test: the multi-rank path of if_fir_mc_* EXECUTED on one GPU — ranks as threads, tests/c/fake_rccl.cpp as the
transport (RCCL refuses two ranks on one device).  See tests/mc_threads_check.py."""
import os
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fake_rccl(gpu_ok):
    tmp = tempfile.mkdtemp(prefix="fake_rccl_")
    so = os.path.join(tmp, "libfake_rccl.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-fPIC", "-shared", "-std=c++17", "-x", "hip",
                           os.path.join(ROOT, "tests", "c", "fake_rccl.cpp"), "-o", so])
    return so


@pytest.mark.parametrize("world,channels", [(2, 3), (3, 5), (4, 4), (4, 2), (4, 8)])   # (4, 8): BASELINE configs[3]'s 8 channels x 255 taps
def test_ranks_as_threads_match_single_channel_contexts(fake_rccl, world, channels):
    run = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "mc_threads_check.py"), fake_rccl, str(world), str(channels)],
                         capture_output=True, text=True, timeout=400)
    assert run.returncode == 0, run.stdout[-3000:] + run.stderr[-3000:]
    assert "bit-identical" in run.stdout


@pytest.mark.parametrize("d", [2, 3, 4, 6, 7, 8, 12, 16, 24, 32])
def test_ranks_as_threads_off_phase_chunks(fake_rccl, d):
    """call lengths that leave the decimation phase != 0, chunked (215040-sample chunks, two slots of staging
    per owned channel on the ranks other than 0): every channel bit-identical to an unchunked single-channel context."""
    run = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "mc_threads_check.py"), fake_rccl, "3", "4", "offphase%d" % d],
                         capture_output=True, text=True, timeout=400)
    assert run.returncode == 0, run.stdout[-3000:] + run.stderr[-3000:]
    assert "bit-identical" in run.stdout


def test_remote_filter_failure_reaches_the_root_and_nobody_hangs(fake_rccl):
    """Rank 1's filter is made to fail (test hook, IF_FIR_DEBUG) in the middle of a chunked call: the protocol completes
    on every rank, rank 1 reports its error, the root reports "rank 1 reported a filter failure" (status word), rank 2
    succeeds; after if_fir_mc_reset on every rank the same contexts produce bit-identical outputs again."""
    run = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "mc_threads_check.py"), fake_rccl, "3", "5", "inject"],
                         capture_output=True, text=True, timeout=400)
    assert run.returncode == 0, run.stdout[-3000:] + run.stderr[-3000:]
    assert "injected failure" in run.stdout and "bit-identical" in run.stdout


def test_remote_block_queue_fault_reaches_the_root(fake_rccl):
    """an expired bounded wait of the overlap-save kernel's block queue leaves blocks unwritten and exists on the
    device only (a counter).  Through the multi-channel front it must still fail the call on the owner (its fault counters
    are read after the final waits) AND on the root (the owner's status word is formed on the device behind its last filter,
    from those counters) -- not gather incomplete outputs with status 0.  Development launch 512 on rank 1's channel."""
    run = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "mc_threads_check.py"), fake_rccl, "3", "5", "qfault"],
                         capture_output=True, text=True, timeout=400)
    assert run.returncode == 0, run.stdout[-3000:] + run.stderr[-3000:]
    assert "bounded wait" in run.stdout and "rank 1 reported a filter failure" in run.stdout and "bit-identical" in run.stdout


def test_asynchronous_rccl_error_is_reported_not_waited_for(fake_rccl):
    """the final waits of if_fir_mc_process_device poll ncclCommGetAsyncError instead of blocking.  The stand-in
    transport makes rank 1's communicator report an error: that rank fails the call with the message, aborts its communicator
    and refuses further calls; the other ranks complete."""
    run = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "mc_threads_check.py"), fake_rccl, "3", "4", "asyncerr"],
                         capture_output=True, text=True, timeout=400)
    assert run.returncode == 0, run.stdout[-3000:] + run.stderr[-3000:]
    assert "communicator aborted, the others completed" in run.stdout
