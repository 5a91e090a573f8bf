#!/usr/bin/env python3
"""mc_selfcheck.py — exercises if_fir_mc_* across ranks on a multi-GPU node (on a one-GPU box: IF_FIR_MC_LOOPBACK=1 python
tools/mc_selfcheck.py runs the same protocol over the real librccl with both ranks played by one process).

launch: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        tools/mc_selfcheck.py [channels] [log2 samples]
Rank 0 synthesises every channel on its GPU, all ranks call if_fir_mc_process_device() twice (streaming state per
channel), rank 0 compares every channel with a single-channel context run locally on the same input and prints the
scatter+filter+gather time.  The RCCL bootstrap id travels through torch.distributed (any other transport would do).

Runs unattended and ends with ONE JSON line on stdout (rank 0; everything else goes to stderr):
  {"mc_selfcheck": <bench.multi_gpu_check_record: per-rank ms of both calls, bytes through ncclSend/ncclRecv per rank,
                    bit-identity per channel, ok>}
exit code 0 iff every channel was bit-identical on both calls.  On the first 8-GPU node:
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29544 tools/mc_selfcheck.py 8 28
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import __graft_entry__ as g  # noqa: E402
import bench  # noqa: E402
import json  # noqa: E402


def main():
    channels = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    n = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 24)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # IF_FIR_MC_SAME_DEVICE=1: every rank on GPU 0 and the bootstrap id over gloo -- an attempt to run the rank-to-rank
    # path on a one-GPU box (RCCL may refuse two ranks on one device; then this prints its error and exits 2)
    same_device = os.environ.get("IF_FIR_MC_SAME_DEVICE") == "1"
    if same_device:
        local = 0
    torch.cuda.set_device(local)
    fir = g.load_pkg().if_fir
    uid = None
    if world > 1:
        if same_device:
            dist.init_process_group("gloo")
            idt = torch.zeros(fir.MC_ID_BYTES, dtype=torch.uint8)
            if rank == 0:
                idt = torch.frombuffer(bytearray(fir.mc_unique_id()), dtype=torch.uint8).clone()
            dist.broadcast(idt, 0)
            uid = bytes(idt.numpy().tobytes())
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            idt = torch.zeros(fir.MC_ID_BYTES, dtype=torch.uint8, device="cuda")
            if rank == 0:
                idt = torch.frombuffer(bytearray(fir.mc_unique_id()), dtype=torch.uint8).cuda()
            dist.broadcast(idt, 0)
            uid = bytes(idt.cpu().numpy().tobytes())
    d, t = 4, 255
    taps = np.stack([fir.bpf_design(t, 0.02 + 0.05 * c, 0.06 + 0.05 * c) for c in range(channels)])
    # IF_FIR_MC_LOOPBACK=N with one rank (development library): this process plays all ranks of an N-rank world over a
    # one-rank communicator of the real librccl -- the protocol end to end on a one-GPU box, only the wire missing
    loopback = world == 1 and os.environ.get("IF_FIR_MC_LOOPBACK", "0") not in ("", "0")
    try:
        mc_ctx = fir.IfFirMc(taps, d, n, device=local, rank=rank, world=world, unique_id=uid, dev=loopback)
    except fir.IfFirError as e:
        print("rank %d: if_fir_mc_init failed: %s" % (rank, e), file=sys.stderr, flush=True)
        sys.exit(2)
    with mc_ctx as mc:
        ins = outs = None
        if rank == 0:
            with fir.IfFir(taps[0], d, 0, device=local) as f:
                ins = [torch.empty(2 * n, dtype=torch.float32, device="cuda") for _ in range(channels)]
                for c in range(channels):
                    f.synth_device(ins[c].data_ptr(), 0, n, c)
                f.synchronize()
                m = f.out_count(n)
            outs = [torch.zeros(2 * m, dtype=torch.float32, device="cuda") for _ in range(channels)]
            torch.cuda.synchronize()
        ok = True
        per_rank, identical_all = [], None
        for call in range(2):
            if world > 1:
                dist.barrier()
            t0 = time.perf_counter()
            got = mc.process_device([x.data_ptr() for x in ins] if rank == 0 else None,
                                    [y.data_ptr() for y in outs] if rank == 0 else None, n)
            dt = time.perf_counter() - t0
            if world > 1:
                t_all = [torch.zeros(1, dtype=torch.float64, device="cuda") for _ in range(world)]
                dist.all_gather(t_all, torch.tensor([dt * 1e3], dtype=torch.float64, device="cuda"))
                per_rank.append([float(v[0]) for v in t_all])
            else:
                per_rank.append([dt * 1e3])
            if rank == 0:
                assert got == m
                identical = []
                for c in range(channels):
                    with fir.IfFir(taps[c], d, 0, device=local) as f:
                        ref = torch.empty_like(outs[c])
                        for _ in range(call + 1):          # same stream state as channel c after `call` earlier calls
                            f.process_device(ins[c].data_ptr(), ref.data_ptr(), n)
                        f.synchronize()
                    same = bool(torch.equal(ref, outs[c]))
                    identical.append(same)
                    ok = ok and same
                    if not same:
                        print("call %d channel %d (rank %d): MISMATCH max |diff| %g" %
                              (call, c, fir.mc_owner(c, world), (ref - outs[c]).abs().max().item()), file=sys.stderr)
                identical_all = identical if identical_all is None else [a and b for a, b in zip(identical_all, identical)]
                print("call %d: %d channels x 2^%d samples over %d ranks%s: %.2f ms (%.1f GS/s end to end) %s" %
                      (call, channels, int(np.log2(n)), world, " (LOOPBACK: %s virtual ranks, real librccl)" % os.environ["IF_FIR_MC_LOOPBACK"] if loopback else "",
                       dt * 1e3, channels * n / dt / 1e9, "OK" if ok else "FAIL"), file=sys.stderr)
        if rank == 0:
            vworld = int(os.environ["IF_FIR_MC_LOOPBACK"]) if loopback else world
            vworld = 2 if loopback and vworld == 1 else vworld
            rec = bench.multi_gpu_check_record(vworld, channels, n, t, d,
                                               "rccl loopback: one process plays all %d ranks over a one-rank communicator" % vworld
                                               if loopback else "rccl (ncclSend/ncclRecv inside libif_fir.so)", per_rank, identical_all)
            sys.stdout.write(json.dumps({"mc_selfcheck": rec}) + "\n")
            sys.stdout.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
