"""CPU tests of the multi-channel front's transfer plan (if_fir_mc_debug_plan, the function if_fir_mc_process_device
executes): every send has exactly one matching receive of equal size, posted in the same group and in the same order
between the two ranks; the bytes add up to whole channels; and a replay of all ranks' transfer streams never blocks."""
import itertools

import pytest

SEND, RECV = 0, 1
SCATTER, GATHER, STATUS = 0, 1, 2


def all_plans(fir, world, channels, samples, in_bytes, decim, consumed, chunk):
    return [fir.mc_debug_plan(world, channels, r, samples, in_bytes, decim, consumed, chunk) for r in range(world)]


def out_count(consumed, n, d):
    n0 = (d - consumed % d) % d
    return (n - n0 + d - 1) // d if n > n0 else 0


CASES = [(s, ch, d, cons) for s, ch, d, cons in [
    (1, 0, 1, 0), (500_000, 215_040, 4, 0), (500_000, 215_040, 4, 3), (645_120, 215_040, 1, 0), (430_081, 215_040, 3, 7),
    (1 << 22, 0, 4, 0), (3, 215_040, 4, 2), (1_000_000, 430_080, 8, 5)]]


@pytest.mark.parametrize("world", [2, 3, 4, 8])
@pytest.mark.parametrize("channels", [1, 3, 8, 13])
def test_every_send_has_one_matching_recv_in_the_same_group_order(fir, world, channels):
    for (samples, chunk, decim, consumed), in_bytes in itertools.product(CASES, (8, 4)):
        plans = all_plans(fir, world, channels, samples, in_bytes, decim, consumed, chunk)
        for r, plan in enumerate(plans):
            groups = [o["group"] for o in plan]
            assert groups == sorted(groups), "a rank posts its groups in increasing order"
            for o in plan:
                assert o["peer"] != r and o["peer"] < world and o["bytes"] > 0
                assert (o["peer"] == 0) != (r == 0), "all traffic is root <-> owner"
        for a, b in itertools.permutations(range(world), 2):
            sends = [o for o in plans[a] if o["kind"] == SEND and o["peer"] == b]
            recvs = [o for o in plans[b] if o["kind"] == RECV and o["peer"] == a]
            assert len(sends) == len(recvs), (world, channels, a, b)
            for s, rv in zip(sends, recvs):   # RCCL matches the operations between two ranks in posting order
                assert (s["group"], s["phase"], s["channel"], s["chunk"], s["bytes"]) == \
                       (rv["group"], rv["phase"], rv["channel"], rv["chunk"], rv["bytes"])
                if s["phase"] != STATUS:
                    assert s["offset"] == rv["offset"]          # same piece of the channel on both sides
        # volume: every remote channel's whole input goes out once, its whole output comes back once
        root = plans[0]
        for c in range(channels):
            if c % world == 0:
                assert not [o for o in root if o["channel"] == c and o["phase"] != STATUS]
                continue
            sc = [o for o in root if o["channel"] == c and o["phase"] == SCATTER]
            ga = [o for o in root if o["channel"] == c and o["phase"] == GATHER]
            assert all(o["kind"] == SEND and o["peer"] == c % world for o in sc)
            assert all(o["kind"] == RECV and o["peer"] == c % world for o in ga)
            assert sum(o["bytes"] for o in sc) == samples * in_bytes
            assert sum(o["bytes"] for o in ga) == 8 * out_count(consumed, samples, decim)
            for ops, unit in ((sc, in_bytes), (ga, 8)):     # contiguous, in order
                pos = 0
                for o in ops:
                    assert o["offset"] == pos and o["bytes"] % unit == 0
                    pos += o["bytes"]
        owners = {c % world for c in range(channels)} - {0}
        assert {o["peer"] for o in root if o["phase"] == STATUS} == owners


@pytest.mark.parametrize("world,channels", [(2, 3), (3, 8), (4, 13), (8, 8), (8, 3)])
def test_replay_of_all_ranks_never_blocks(fir, world, channels):
    """Every rank executes its groups in order on one stream; a group completes when each of its operations has its
    counterpart in the group the peer is currently executing.  The replay must drain every rank."""
    for samples, chunk, decim, consumed in CASES:
        plans = all_plans(fir, world, channels, samples, 8, decim, consumed, chunk)
        queues = []
        for plan in plans:
            q = []
            for o in plan:
                if not q or q[-1][0]["group"] != o["group"]:
                    q.append([])
                q[-1].append(o)
            queues.append(q)
        pos = [0] * world

        def current(r):
            return queues[r][pos[r]] if pos[r] < len(queues[r]) else None

        def matched(r):
            for o in current(r):
                peer_group = current(o["peer"])
                if peer_group is None:
                    return False
                want = (1 - o["kind"], r, o["phase"], o["channel"], o["chunk"], o["bytes"])
                if not any((p["kind"], p["peer"], p["phase"], p["channel"], p["chunk"], p["bytes"]) == want for p in peer_group):
                    return False
            return True

        for _ in range(10_000):
            ready = [r for r in range(world) if current(r) is not None and matched(r)]
            if not ready:
                break
            # a group between two ranks completes on both at once; ranks whose whole group is matched move on
            for r in ready:
                pos[r] += 1
        assert all(current(r) is None for r in range(world)), (world, channels, samples, chunk, pos)


def test_plan_rejects_bad_arguments_and_single_rank_moves_nothing(fir):
    assert fir.mc_debug_plan(1, 5, 0, 1 << 20) == []
    assert fir.mc_debug_plan(4, 2, 3, 1 << 20) == []          # a rank without channels takes no part
    assert fir.dev_lib().if_fir_mc_debug_plan(0, 1, 0, 10, 8, 255, 1, 0, 0, None, 0) == 0
    assert fir.dev_lib().if_fir_mc_debug_plan(2, 1, 2, 10, 8, 255, 1, 0, 0, None, 0) == 0
    assert fir.dev_lib().if_fir_mc_debug_plan(2, 1, 0, 10, 5, 255, 1, 0, 0, None, 0) == 0
    assert fir.dev_lib().if_fir_mc_debug_plan(2, 1, 0, 10, 8, 0, 1, 0, 0, None, 0) == 0


def test_stand_in_transport_builds():
    """tests/c/fake_rccl.cpp (the in-process transport the GPU tests load through IF_FIR_RCCL_LIBRARY) cross-compiles for
    gfx950 and exports the eight entry points if_fir_mc.cpp resolves."""
    import os
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(tempfile.mkdtemp(prefix="fake_rccl_"), "libfake_rccl.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-fPIC", "-shared", "-std=c++17", "-x", "hip",
                           os.path.join(root, "tests", "c", "fake_rccl.cpp"), "-o", so])
    syms = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True).stdout
    for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclCommAbort", "ncclSend", "ncclRecv",
                 "ncclGroupStart", "ncclGroupEnd", "ncclGetErrorString", "ncclCommGetAsyncError"):
        assert (" T " + name) in syms, name


def has_decimating_tail(taps, decim):
    """if_fir::fft_tail: every even decimation (the decimate-by-4 / -by-2 tail keeping every sub-th output), any tap count;
    round 4: and the odd-decimation kernel (decimation 3, 9, 15, ... with at most 767 taps) -- block grids anchored at the call's
    first output"""
    return taps <= 4096 and (decim % 2 == 0 or (decim % 3 == 0 and taps <= 767))


def block_advance(taps, decim):
    """if_fir::fft_block_advance: new input samples per block of the overlap-save kernel for this filter"""
    if taps > 3073:
        return 2048                                   # two partitions, each on the 32-row kernel
    if decim % 2 == 1 and decim % 3 == 0 and (taps - 1 + 2 + 2) // 3 <= 256:
        need = (taps - 1 + 2 + 2) // 3                # round 4: the odd-decimation kernel, blocks of 3 x 1024 samples
        return 3 * (1024 - 64 * (2 if need <= 128 else 4))
    rows = 4 if taps - 1 <= 256 else 8 if taps - 1 <= 512 else 16 if taps - 1 <= 1024 else 32 if taps - 1 <= 2048 else 48
    return 4096 - 64 * rows


def effective_chunk(request, taps, decim):
    """mc_effective_chunk: the multiple of unit = lcm(block advance, 2 D) nearest to the request (at least one unit)"""
    import math
    adv = block_advance(taps, decim)
    unit = adv * (2 * decim) // math.gcd(adv, 2 * decim)
    return max(1, (request + unit // 2) // unit) * unit, unit


@pytest.mark.parametrize("decim", [1, 2, 3, 4, 6, 7, 8, 9, 11, 12, 15, 16, 17, 24, 32, 48, 60, 61, 63, 64])
def test_chunk_table_keeps_output_pieces_aligned_and_chunks_on_the_block_grid(fir, decim):
    """ADVICE r2 / r3, VERDICT r2 #8 / r3 #1: the effective chunk is the multiple of the context's unit -- lcm(block advance of
    ITS filter, 2 D) -- nearest to the request, so every chunk produces an even number of outputs at any phase (the gather
    pieces start 16-byte aligned) and stays close to the request for every decimation (round 3 took lcm(request, 2 D): 11 .. 61
    times the request for decimations with a prime factor the request lacked); where the kernel's block grid follows the
    decimation phase (the decimating tails: every even D) the first chunk of an off-phase call absorbs the phase: every later chunk
    starts on phase 0, a whole number of block advances after the call's first output."""
    request = 215_040
    for consumed in (0, 1, 5, decim - 1, 3 * request + 5):
        for taps in (127, 255, 1023, 3075):
            eff, unit = effective_chunk(request, taps, decim)
            assert eff % block_advance(taps, decim) == 0 and eff % (2 * decim) == 0
            assert abs(eff - request) <= unit // 2 or eff == unit
            assert eff <= 2 * request, (decim, taps, eff)        # never blown up
            n0 = (decim - consumed % decim) % decim
            shift = n0 if has_decimating_tail(taps, decim) else 0
            samples = 3 * eff + eff // 2 + 3
            unit = request                                        # (the request handed to the plan below)
            plan = fir.mc_debug_plan(2, 2, 0, samples, 8, decim, consumed, unit, taps=taps)
            sc = [o for o in plan if o["phase"] == SCATTER]
            ga = [o for o in plan if o["phase"] == GATHER]
            assert len(sc) == 4 and len(ga) == 4, (decim, consumed, taps, len(sc))
            assert [o["bytes"] // 8 for o in sc] == [eff + shift, eff, eff, samples - 3 * eff - shift]
            for o in ga:
                assert o["offset"] % 16 == 0, (decim, consumed, taps, o)
            for o in sc[1:]:
                first = o["offset"] // 8
                assert (first - shift) % eff == 0
                if shift:
                    assert (consumed + first) % decim == 0      # later chunks start on phase 0
            # int16 input: the same chunk table, half the bytes
            plan4 = fir.mc_debug_plan(2, 2, 0, samples, 4, decim, consumed, unit, taps=taps)
            assert [o["bytes"] // 4 for o in plan4 if o["phase"] == SCATTER] == [o["bytes"] // 8 for o in sc]
