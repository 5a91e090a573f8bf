#!/usr/bin/env python3
"""mc_threads_check.py <fake_rccl.so> <world> <channels> — runs the multi-rank path of if_fir_mc_* on ONE GPU with the
ranks as threads of this process and tests/c/fake_rccl.cpp as the transport (IF_FIR_RCCL_LIBRARY).  Everything of
if_fir_mc_process_device executes as on a multi-GPU node — the transfer plan, chunking, the transfer and filter streams,
their events, the status word — except that a "link" is a device-to-device copy.  Rank 0 checks every channel's output
bit for bit against a single-channel context fed the same calls.  Started as a subprocess by tests/test_mc_threads.py
(the library caches the RCCL entry points per process)."""
import os
import sys
import threading

fake, world, channels = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
mode = sys.argv[4] if len(sys.argv) > 4 else ""
qfault = mode == "qfault"       # rank 1's first channel's block queue counts a fault during the first call (development launch 512)
inject = mode == "inject" or qfault  # rank 1's first channel fails its first filter call (development library: test hook)
offphase = mode.startswith("offphase")   # "offphase<D>": call lengths that leave the decimation phase != 0
asyncerr = mode == "asyncerr"   # rank 1's communicator reports an asynchronous error (stand-in transport hook)
if asyncerr:
    os.environ["FAKE_RCCL_ASYNC_ERROR_RANK"] = "1"
os.environ["IF_FIR_RCCL_LIBRARY"] = fake
os.environ["IF_FIR_DEBUG"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import __graft_entry__ as g  # noqa: E402

fir = g.load_pkg().if_fir
torch.cuda.set_device(0)
d, t = (int(mode[8:]) if offphase and mode[8:] else 4), 255
# two calls: streaming state per channel.  Default: phase 0 at every chunk boundary; offphase: the second call starts at
# phase 1 and both calls are cut into chunks (the first chunk of an off-phase call absorbs the phase when D = 4)
calls = [700_001, 500_003] if offphase else [700_000, 300_004]
nmax = max(calls)
taps = np.stack([fir.bpf_design(t, 0.02 + 0.04 * c, 0.05 + 0.04 * c) for c in range(channels)])
uid = fir.mc_unique_id()
assert uid.startswith(b"fake-rccl-world-"), "the stand-in transport was not loaded"
with fir.IfFir(taps[0], d, 0) as f:
    ins = [[torch.empty(2 * n, dtype=torch.float32, device="cuda") for _ in range(channels)] for n in calls]
    first = 0
    for k, n in enumerate(calls):
        for c in range(channels):
            f.synth_device(ins[k][c].data_ptr(), first, n, c)
        first += n
    f.synchronize()
outs = [[torch.zeros(2 * ((n + d - 1) // d) + 8, dtype=torch.float32, device="cuda") for _ in range(channels)] for n in calls]
torch.cuda.synchronize()
errors, counts, injected = [], {}, {}
barrier = threading.Barrier(world)


def rank_main(rank):
    try:
        torch.cuda.set_device(0)
        with fir.IfFirMc(taps, d, nmax, device=0, rank=rank, world=world, unique_id=uid, dev=inject) as mc:
            mc.set_chunk_samples(fir.MC_CHUNK_UNIT)        # 215040 samples: four chunks in the first call, two in the second
            if inject:
                # an owner's filter fails in the middle of the protocol: nobody may hang, the owner and the root must both
                # report it, the others succeed; after a reset on every rank the front works again
                if rank == 1:
                    h = mc.channel_ctx(1)
                    # 4000: the next filter call fails on the host before anything is launched; 1000512 (qfault): the launches
                    # themselves are accepted, the waves of workgroup 0 count a queue fault and leave their blocks unwritten --
                    # the failure exists on the DEVICE only and must still reach the owner's return value and the root's
                    assert h and fir.dev_lib().if_fir_set_tuning(h, 1000000 + 512 if qfault else 4000)
                barrier.wait()
                try:
                    mc.process_device([x.data_ptr() for x in ins[0]] if rank == 0 else None,
                                      [y.data_ptr() for y in outs[0]] if rank == 0 else None, calls[0])
                    injected[rank] = "ok"
                except fir.IfFirError as e:
                    injected[rank] = str(e)
                barrier.wait()
                if rank == 1 and qfault:
                    assert fir.dev_lib().if_fir_set_tuning(mc.channel_ctx(1), 0)
                mc.reset()
                for k in range(len(calls)):
                    for y in outs[k]:
                        y.zero_() if rank == 0 else None
                torch.cuda.synchronize()
            if asyncerr:
                # the library's waits poll ncclCommGetAsyncError: the rank whose communicator reports an error fails the
                # call with that message, aborts the communicator and refuses further calls; the others complete
                barrier.wait()
                try:
                    mc.process_device([x.data_ptr() for x in ins[0]] if rank == 0 else None,
                                      [y.data_ptr() for y in outs[0]] if rank == 0 else None, calls[0])
                    injected[rank] = "ok"
                except fir.IfFirError as e:
                    injected[rank] = str(e)
                if rank == 1:
                    try:
                        mc.process_device(None, None, calls[0])
                        injected["again"] = "ok"
                    except fir.IfFirError as e:
                        injected["again"] = str(e)
                return
            for k, n in enumerate(calls):
                barrier.wait()
                m = mc.process_device([x.data_ptr() for x in ins[k]] if rank == 0 else None,
                                      [y.data_ptr() for y in outs[k]] if rank == 0 else None, n)
                counts[(rank, k)] = m
    except Exception as e:   # noqa: BLE001
        errors.append("rank %d: %r" % (rank, e))
        try:
            barrier.abort()
        except Exception:   # noqa: BLE001
            pass


threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
for th in threads:
    th.start()
for th in threads:
    th.join(timeout=120)
if any(th.is_alive() for th in threads):
    print("FAIL: a rank did not return within 120 s")
    os._exit(3)
if errors:
    print("FAIL:", "; ".join(errors))
    sys.exit(1)
ok = True
if asyncerr:
    print("asynchronous error:", injected)
    good = ("asynchronous error" in injected.get(1, "") and "aborted" in injected.get("again", "")
            and all(injected.get(r) == "ok" for r in range(world) if r != 1))
    print("asynchronous error reported by the rank that saw it, communicator aborted, the others completed" if good else "FAIL")
    sys.exit(0 if good else 1)
if inject:
    print("injected failure:", injected)
    ok = (("bounded wait" if qfault else "injected failure") in injected.get(1, "") and "rank 1 reported a filter failure" in injected.get(0, "")
          and all(injected.get(r) == "ok" for r in range(2, world)))
for c in range(channels):
    with fir.IfFir(taps[c], d, 0) as f:
        for k, n in enumerate(calls):
            ref = torch.empty(2 * f.out_count(n), dtype=torch.float32, device="cuda")
            m = f.process_device(ins[k][c].data_ptr(), ref.data_ptr(), n)
            f.synchronize()
            got = outs[k][c]
            same = bool(torch.equal(ref, got[:2 * m])) and bool((got[2 * m:] == 0).all())
            if not same or any(counts[(r, k)] != m for r in range(world)):
                ok = False
                print("channel %d (rank %d) call %d: MISMATCH (max |diff| %g, counts %s)" %
                      (c, c % world, k, (ref - got[:2 * m]).abs().max().item(), [counts[(r, k)] for r in range(world)]))
print("%d ranks as threads, %d channels, calls %s, chunks of %d samples: %s" %
      (world, channels, calls, fir.MC_CHUNK_UNIT, "all channels bit-identical to single-channel contexts" if ok else "FAIL"))
sys.exit(0 if ok else 1)
