#!/usr/bin/env python3
"""bench.py — headline benchmark of the IF-chain FIR path (BASELINE.json metric) on N MI355X of one node.

A "step" is one pass of the hot path (if_fir_process_device through the C-ABI of libif_fir.so) over one batch of
device-resident synthetic IQ samples.  Default workload = BASELINE.json configs[2], the configuration the metric is
quoted on: 255-tap complex-IQ FIR + decimate-by-4 polyphase, one channel of 2^28 samples per GPU.  With --gpus N the
driver launches one rank per GPU (torch.distributed, backend nccl = RCCL); channels are independent, so ranks share
no data-path collective and scaling is weak (one channel per GPU).  `--scatter` additionally times the RCCL
scatter/gather of whole channels from rank 0 (reported in "extra", never in `value`).

Prints ONE JSON line on rank 0.  The oracle (oracle/) is used only for the cpu_baseline leg and a parity spot check.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md chip table)
BARRIER_COVER_STEPS = 40   # multi-rank runs: untimed passes of the step queued in front of the opening barrier (see main)
VALU_PEAK_TFLOPS = 157.3   # FP32 vector peak, packed FMA counted

WORKLOADS = {
    # name: (taps, decimation, log2 samples, description)
    "fir255_dec4_2p28": (255, 4, 28, "255-tap complex-IQ FIR + decimate-by-4 polyphase, one channel per GPU, 2^28 IQ samples (BASELINE configs[2])"),
    "fir127_2p26": (127, 1, 26, "127-tap complex-IQ FIR, one channel per GPU, 2^26 IQ samples (BASELINE configs[1])"),
    "fir255_2p28": (255, 1, 28, "255-tap complex-IQ FIR, no decimation, one channel per GPU, 2^28 IQ samples"),
    "fir255_dec4_2p24": (255, 4, 24, "255-tap + decimate-by-4, 2^24 IQ samples (quick check size)"),
    "fir255_dec4_i16_2p28": (255, 4, 28, "255-tap FIR + decimate-by-4, int16 IQ input front-end (SURVEY §8f-1), 2^28 samples"),
    "fir255_dec4_nco_2p28": (255, 4, 28, "NCO mix (0.19995 cycles/sample) fused into the 255-tap FIR + decimate-by-4 (SURVEY §8f-1), 2^28 samples"),
    "fir1023_dec8_2p28": (1023, 8, 28, "1023-tap FIR, decimate-by-8 (overlap-save at full rate + selecting store), 2^28 samples"),
    "fir255_dec2_2p28": (255, 2, 28, "255-tap FIR, decimate-by-2 (overlap-save at full rate + selecting store), 2^28 samples"),
    "fir2047_dec8_2p26": (2047, 8, 26, "2047-tap FIR, decimate-by-8, 2^26 IQ samples (32-row overlap, selecting store)"),
    "fir255_dec3_2p28": (255, 3, 28, "255-tap FIR, decimate-by-3 (round 4: blocks of 3 x 1024 samples, three forward 1024-point transforms + one inverse), 2^28 samples"),
    "fir255_dec9_2p28": (255, 9, 28, "255-tap FIR, decimate-by-9 (the decimate-by-3 kernel keeping every 3rd output), 2^28 samples"),
    "fir511_dec3_2p28": (511, 3, 28, "511-tap FIR, decimate-by-3 (4 of 16 output rows of a block dropped), 2^28 samples"),
    "fir255_dec6_2p28": (255, 6, 28, "255-tap FIR, decimate-by-6 (the decimate-by-2 tail keeping every third output), 2^28 samples"),
    "fir255_dec12_2p28": (255, 12, 28, "255-tap FIR, decimate-by-12 (the decimate-by-4 tail keeping every third output), 2^28 samples"),
    "fir255_dec5_2p28": (255, 5, 28, "255-tap FIR, decimate-by-5 (full-rate pipeline + selecting store: no decimating tail for 5, 7, 11, ...), 2^28 samples"),
    "fir255_dec7_2p28": (255, 7, 28, "255-tap FIR, decimate-by-7 (selecting store), 2^28 samples"),
    "fir255_dec25_2p28": (255, 25, 28, "255-tap FIR, decimate-by-25 (selecting store), 2^28 samples"),
    "fir1023_dec3_2p28": (1023, 3, 28, "1023-tap FIR, decimate-by-3 (selecting store: more than 767 taps), 2^28 samples"),
    "fir1023_2p28": (1023, 1, 28, "1023-tap complex-IQ FIR, one channel per GPU, 2^28 IQ samples (BASELINE configs[4])"),
}


def algorithmic_bytes_per_sample(decim, in_bytes=8.0):
    return in_bytes + 8.0 / decim     # read 8 B (4 B for int16 input) per input sample, write 8/D


def whole_output_check(fir, f, taps, decim, x, y, n, i16, stream, device, calls, names, nco=0.0):
    """Every output sample of the timed context's last call against another kernel family: the reference context is
    primed with the tail of x (the timed stream's history: x is fed again every step, n is a multiple of the
    decimation) and filters x once.  Both sides are within 1e-6 of the float64 result, so they agree within 2e-6."""
    run_b = f.get_backend()
    direct_ok = (not i16) and (not nco) and taps.size in (127, 255) and decim in (1, 4)
    if run_b != fir.BACKEND_HIP_DIRECT and direct_ok:
        ref_b = fir.BACKEND_HIP_DIRECT
    elif run_b != fir.BACKEND_HIP_GENERIC:
        ref_b = fir.BACKEND_HIP_GENERIC
    else:
        ref_b = fir.BACKEND_HIP_FFT if (decim in (1, 4) and taps.size <= 1025) else fir.BACKEND_HIP_TAPSPLIT
    tail = 8192
    if calls < 1 or n % decim or tail % decim or n < tail:
        return {"ok": True, "skipped": "stream state not reproducible for this configuration"}
    esz = 4 if i16 else 8
    with fir.IfFir(taps, decim, 0, device=device) as fr:
        fr.set_stream(stream.cuda_stream)
        if i16:
            fr.set_input_format(fir.INPUT_I16)
        if nco:
            fr.set_nco(nco)      # bench frequencies satisfy P * tail = 0 (mod 2^32): same phase as the timed stream
        fr.set_backend(ref_b)
        yref = torch.empty_like(y)
        if calls > 1:
            fr.process_device(x.data_ptr() + esz * (n - tail), yref.data_ptr(), tail)
        m = fr.process_device(x.data_ptr(), yref.data_ptr(), n)
        fr.synchronize()
        torch.cuda.synchronize()
        scale = float(yref.abs().max().item())
        err = float((y - yref).abs().max().item())
        nz = float((yref != 0).double().mean().item())
    return {"ok": bool(scale > 0 and err <= 2e-6 * scale and 2 * m == y.numel()), "rel_max_diff": err / max(scale, 1e-30),
            "bound": 2e-6, "against": names[ref_b], "outputs": int(m), "nonzero_fraction": nz}


def algorithmic_flops_per_sample(taps, decim):
    return 4.0 * taps / decim         # real taps x complex data: 2 FMA = 4 flop per tap per OUTPUT sample


def cpu_baseline(taps_arr, decim, budget_s=10.0):
    """Time the oracle's float32 OpenMP direct form on the host cores (kind "port": no reference CPU code exists)."""
    oracle = graft.load_oracle()
    handle = None
    try:  # rebuild with -march=native for this host into a temp dir (the shipped .so is x86-64-v3)
        out = os.path.join(tempfile.mkdtemp(prefix="if_fir_oracle_"), "libif_fir_oracle_native.so")
        subprocess.check_call(["gcc", "-O3", "-march=native", "-fopenmp", "-fPIC", "-std=gnu11", "-shared", "-o", out,
                               os.path.join(ROOT, "oracle", "if_fir_oracle.c"), "-lm"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        handle = oracle.lib(out)
    except Exception:
        handle = None
    lib = handle or oracle.lib()
    threads = min(oracle.max_threads(), 16)   # a 1-GPU box's CPU share is 16 cores
    n = 1 << 26
    t = taps_arr.size
    buf = np.zeros(2 * (n + t - 1), dtype=np.float32)      # T-1 zero history in front of the stream (phase 0)
    buf[2 * (t - 1):] = oracle.synth_iq(n)
    y = np.empty(2 * ((n + decim - 1) // decim), dtype=np.float32)
    import ctypes
    f32p = ctypes.POINTER(ctypes.c_float)
    args = (taps_arr.ctypes.data_as(f32p), t, decim, buf.ctypes.data_as(f32p), n, y.ctypes.data_as(f32p), threads)
    lib.oracle_fir_c64_f32_omp(*args)                      # warm-up (page faults, thread pool)
    times, spent = [], 0.0
    while spent < budget_s and len(times) < 4096:   # ~10 s of wall on all `threads` cores
        t0 = time.perf_counter()
        lib.oracle_fir_c64_f32_omp(*args)
        times.append(time.perf_counter() - t0)
        spent += times[-1]
    kernel_dt, reps = min(times), len(times)
    n1 = 1 << 22                                             # single-thread figure on a shorter sample (~1 s)
    args1 = (taps_arr.ctypes.data_as(f32p), t, decim, buf.ctypes.data_as(f32p), n1, y.ctypes.data_as(f32p), 1)
    t1 = []
    for _ in range(3):
        t0 = time.perf_counter()
        lib.oracle_fir_c64_f32_omp(*args1)
        t1.append(time.perf_counter() - t0)
    return {"value": round(n / kernel_dt / 1e6, 2), "unit": "MSamples/s", "cores": threads, "kind": "port",
            "median": round(n / float(np.median(times)) / 1e6, 2), "cpu_seconds": round(spent, 1),
            "single_thread": round(n1 / min(t1) / 1e6, 2),
            "sample": "2^26 IQ samples of the same synthetic stream, taps=%d decimation=%d, float32 OpenMP direct form "
                      "(oracle/if_fir_oracle.c oracle_fir_c64_f32_omp), best of %d runs, %s build; build-authored CPU "
                      "baseline: the reference has no CPU implementation" %
                      (taps_arr.size, decim, reps, "-march=native" if handle is not None else "x86-64-v3")}


def live_traffic(workload, backend, timeout_s=150.0):
    """HBM bytes per launch of this workload's kernel measured ON THIS BOX, in this run: two rocprofv3 --pmc passes
    (FETCH_SIZE, WRITE_SIZE; each in its own pass, no trace domains) around a 3-step child run of this script.  Must be
    called BEFORE this process touches the GPU (the children are separate processes).  Returns None when the profiler is
    missing, refuses, hangs (bounded by timeout_s per pass) or when this process already runs under a profiler."""
    import csv
    import glob
    import shutil
    if "rocprofiler" in os.environ.get("LD_PRELOAD", "") or os.environ.get("ROCPROFILER_LIBRARY_CTOR") or \
            os.environ.get("IF_FIR_BENCH_NO_PROFILER"):
        return None
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None
    out = tempfile.mkdtemp(prefix="if_fir_traffic_")
    env = dict(os.environ, TMPDIR="/tmp")
    got = {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(out, ctr)
            cmd = [prof, "--pmc", ctr, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
                   "--workload", workload, "--backend", backend, "--traffic-child"]
            try:
                subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s)
            except subprocess.TimeoutExpired:
                return None
            per_kernel = {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r["Counter_Name"] == ctr and "if_fir::fir_" in r["Kernel_Name"]:
                        per_kernel.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
            if not per_kernel:
                return None
            # the filter kernel is the one that moves the bytes; drop the small head-of-stream launch of the child
            name = max(per_kernel, key=lambda k: max(per_kernel[k]))
            vals = [v for v in per_kernel[name] if v > 0.5 * max(per_kernel[name])]
            got[ctr] = (sum(vals) / len(vals), len(vals), name.split("(")[0])
    finally:
        shutil.rmtree(out, ignore_errors=True)
    read_b, write_b = 2.0 * got["FETCH_SIZE"][0] * 1024.0, got["WRITE_SIZE"][0] * 1024.0
    return {"hbm_bytes_per_launch": read_b + write_b, "read_bytes_per_launch": read_b, "write_bytes_per_launch": write_b,
            "launches_averaged": [got["FETCH_SIZE"][1], got["WRITE_SIZE"][1]], "kernel": got["FETCH_SIZE"][2]}


def traffic_child(args):
    """--traffic-child: three passes of the workload's step and nothing else (what live_traffic() profiles)."""
    pkg = graft.load_pkg()
    fir = pkg.if_fir
    taps_n, decim, log2n, _ = WORKLOADS[args.workload]
    n = 1 << log2n
    i16 = "_i16_" in args.workload
    nco = 1638.0 / 8192.0 if "_nco_" in args.workload else 0.0
    torch.cuda.set_device(0)
    backend_ids = {"auto": fir.BACKEND_AUTO, "direct": fir.BACKEND_HIP_DIRECT, "fft": fir.BACKEND_HIP_FFT,
                   "generic": fir.BACKEND_HIP_GENERIC}
    with fir.IfFir(fir.bpf_design(taps_n), decim, 0, device=0, backend=backend_ids[args.backend]) as f:
        if nco:
            f.set_nco(nco)
        x = torch.empty(2 * n, dtype=torch.float32, device="cuda")
        y = torch.empty(2 * f.out_count(n), dtype=torch.float32, device="cuda")
        f.synth_device(x.data_ptr(), 0, n, 0)
        f.synchronize()
        if i16:
            f.set_input_format(fir.INPUT_I16)
            x = torch.clamp(torch.round(x * 16384.0), -32768, 32767).to(torch.int16)
            torch.cuda.synchronize()
        for _ in range(3):
            f.process_device(x.data_ptr(), y.data_ptr(), n)
        f.synchronize()


def free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def self_launch_argv(gpus, argv, port):
    """argv of the child that runs this script as `gpus` ranks (one per GPU) under torch.distributed.run."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(gpus, argv):
    """`python3 bench.py --gpus N` without a launcher: start the N ranks as a CHILD process (never exec: under
    rocprofv3 the preloaded library has already initialised the GPU in this process), relay rank 0's JSON line and
    return the child's exit code.  Nothing in this process has touched the GPU at this point."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = self_launch_argv(gpus, argv, free_port())
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env)
    out = proc.communicate()[0].decode(errors="replace")
    lines = []
    for ln in out.splitlines():
        try:
            if "metric" in json.loads(ln):
                lines.append(ln)
        except ValueError:
            sys.stderr.write(ln + "\n")       # launcher chatter is not part of the one-line contract
    if lines:
        sys.stdout.write(lines[-1] + "\n")
        sys.stdout.flush()
    return proc.returncode if proc.returncode else (0 if lines else 1)


def dry_run(args, rank, world, json_fd):
    """--dry-run: the launch / barrier / MAX-over-ranks / one-JSON-line plumbing on the CPU (gloo), with a timing stub
    in place of the filter (no GPU, no oracle, no filtering: the line says so and carries no roofline)."""
    use_dist = world > 1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    taps_n, decim, log2n, desc = WORKLOADS[args.workload]
    n = 1 << 12
    total_channels = args.channels if args.channels else world
    if total_channels < world:
        raise SystemExit("--channels must be at least the number of GPUs")
    owned = [c for c in range(total_channels) if c % world == rank]

    def step_stub():
        for _ in owned:
            time.sleep(1e-4)

    for _ in range(args.warmup):
        step_stub()
    if use_dist:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step_stub()
    if use_dist:
        dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall = float(t[0])
    extra = {}
    if args.scatter and use_dist:
        extra["scatter_gather"] = dry_run_scatter(rank, world, decim)
    if args.scatter_mc and use_dist:
        extra["scatter_gather_mc"] = dry_run_scatter_mc(rank, world, taps_n, decim)
    if rank == 0:
        line = {"metric": "complex-IQ MSamples/s through %d-tap FIR" % taps_n,
                "value": round(total_channels * n / (wall / args.steps) / 1e6, 3), "unit": "MSamples/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(wall / args.steps * 1e3, 4),
                "higher_is_better": True, "scaling": "strong" if args.channels else "weak", "vs_baseline": None,
                "dtype": "f32", "data": "none", "dry_run": True,
                "config": {"workload": "DRY RUN of the launch path: a sleep stands for the filter (%s)" % desc,
                           "name": args.workload, "channels": total_channels, "channels_on_rank0": len(owned),
                           "parallelism": "channel c on rank c mod %d, no data-path collective" % world}}
        if extra:
            line["extra"] = extra
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


MULTI_GPU_CHECK_KEYS = ("world", "channels", "samples_per_channel", "taps", "decimation", "transport", "calls", "per_rank_ms",
                        "end_to_end_ms", "end_to_end_msamples_per_s", "bytes_sent_per_rank", "bytes_received_per_rank",
                        "bytes_through_send_recv", "channels_identical", "ok")


def mc_transfer_bytes(world, channels, samples, outputs, in_bytes=8):
    """Bytes every rank hands to ncclSend / takes from ncclRecv in ONE call of the multi-channel front (if_fir_mc.cpp's
    plan): channel c != 0 mod world travels root -> owner (samples x in_bytes) and its outputs back (outputs x 8); every
    owner other than the root sends a 4-byte status word."""
    sent, recv = [0] * world, [0] * world
    for c in range(channels):
        o = c % world
        if o == 0:
            continue
        sent[0] += samples * in_bytes
        recv[o] += samples * in_bytes
        sent[o] += outputs * 8
        recv[0] += outputs * 8
    for r in range(1, min(world, channels)):
        sent[r] += 4
        recv[0] += 4
    return sent, recv


def multi_gpu_check_record(world, channels, samples, taps, decimation, transport, per_rank_ms_calls, identical, in_bytes=8,
                           outputs=None):
    """The record a multi-GPU box must produce in one go: per-rank wall times of every call of the multi-channel
    front, the bytes that went through ncclSend / ncclRecv, and per channel whether rank 0 found it bit-identical to a
    single-channel context.  Printed by `bench.py --gpus N --channels C --scatter-mc` (extra.scatter_gather_mc.multi_gpu_check),
    by its --dry-run rehearsal over gloo, and by tools/mc_selfcheck.py (one JSON line)."""
    if outputs is None:   # (a stream position on the decimation grid; off-phase callers pass their count)
        outputs = (samples + decimation - 1) // decimation
    sent, recv = mc_transfer_bytes(world, channels, samples, outputs, in_bytes)
    best = min(max(call) for call in per_rank_ms_calls) if per_rank_ms_calls else None
    rec = {"world": world, "channels": channels, "samples_per_channel": samples, "taps": taps, "decimation": decimation,
           "transport": transport, "calls": len(per_rank_ms_calls),
           "per_rank_ms": [[round(v, 4) for v in call] for call in per_rank_ms_calls],
           "end_to_end_ms": round(best, 4) if best is not None else None,
           "end_to_end_msamples_per_s": round(channels * samples / (best * 1e-3) / 1e6, 1) if best else None,
           "bytes_sent_per_rank": sent, "bytes_received_per_rank": recv, "bytes_through_send_recv": sum(sent),
           "channels_identical": identical,
           "ok": bool(identical) and all(v is True for v in identical)}
    assert tuple(rec) == MULTI_GPU_CHECK_KEYS
    return rec


def dry_run_scatter(rank, world, decim):
    """--dry-run --scatter: the control flow of the real --scatter leg (channel_shard.scatter_channels -> per-channel
    filter -> gather_outputs, barriers in the same places) on CPU tensors over gloo; the filter is a stand-in that keeps
    every decim-th sample, so rank 0 can check that every channel came back from the right owner in the right place."""
    cs = graft.load_pkg().channel_shard
    cpu = torch.device("cpu")
    n = 1 << 12
    m = (n + decim - 1) // decim
    root_inputs = None
    if rank == 0:
        root_inputs = [torch.arange(2 * n, dtype=torch.float32) + 100000.0 * c for c in range(world)]
    dist.barrier()
    mine = cs.scatter_channels(root_inputs, world, n, cpu, root=0)
    dist.barrier()
    outs = {c: xin.view(-1, 2)[::decim].reshape(-1).clone() for c, xin in mine.items()}
    dist.barrier()
    res = cs.gather_outputs(outs, world, m, cpu, root=0)
    dist.barrier()
    ok = None
    if rank == 0:
        ok = all(torch.equal(res[c], root_inputs[c].view(-1, 2)[::decim].reshape(-1)) for c in range(world))
    return {"dry_run": True, "channels": world, "ok": ok,
            "note": "channel_shard scatter -> stand-in filter -> gather over gloo (CPU rehearsal of the --scatter leg)"}


def dry_run_scatter_mc(rank, world, taps_n, decim):
    """--dry-run --scatter-mc: the transfer plan the C front executes (if_fir_mc_debug_plan: chunks, groups, posting
    order) run by REAL processes over gloo: every group is one batch of point-to-point operations, an owner "filters" a
    chunk (stand-in: keeps the samples on the decimation grid) between its scatter and its gather, chunk k lives in slot
    k & 1 of the owners' staging as in if_fir_mc.cpp.  Rank 0 checks every remote channel's output bytes and the status
    words.  An off-phase stream position and three chunks are used on purpose."""
    fir = graft.load_pkg().if_fir
    unit = fir.MC_CHUNK_UNIT
    consumed, samples, channels = 3, 2 * unit + 1001, world + 1
    plan = fir.mc_debug_plan(world, channels, rank, samples, 8, decim, consumed, unit, taps=taps_n)
    n0 = (decim - consumed % decim) % decim
    m_total = (samples - n0 + decim - 1) // decim if samples > n0 else 0
    root = rank == 0

    def pattern(c):   # int64 sample ids as the 8-byte "samples" of channel c
        return torch.arange(samples, dtype=torch.int64) + (c << 40)

    ins = {c: pattern(c) for c in range(channels)} if root else {}
    outs = {c: torch.zeros(m_total, dtype=torch.int64) for c in range(channels)} if root else {}
    slot_in = max([o["bytes"] // 8 for o in plan if o["phase"] == 0] + [1])
    slot_out = max([o["bytes"] // 8 for o in plan if o["phase"] == 1] + [1])
    owned = [c for c in range(channels) if c % world == rank]
    st_in = {c: torch.zeros(2 * slot_in, dtype=torch.int64) for c in owned} if not root else {}
    st_out = {c: torch.zeros(2 * slot_out, dtype=torch.int64) for c in owned} if not root else {}
    status = torch.zeros(1 + world, dtype=torch.int32)
    chunk_first = {}   # chunk -> first input sample (from the scatter offsets this rank sees)
    groups = {}
    for o in plan:
        groups.setdefault(o["group"], []).append(o)
    last_group = max([o["group"] for o in plan] + [-1])
    # every rank walks the same group numbers (a rank without operations in a group skips it)
    gmax = torch.tensor([last_group], dtype=torch.int64)
    dist.all_reduce(gmax, op=dist.ReduceOp.MAX)
    t_walk = time.perf_counter()
    for g in range(int(gmax.item()) + 1):
        ops, after = [], []
        for o in groups.get(g, []):
            k, c, cnt = o["chunk"], o["channel"], o["bytes"] // 8
            if o["phase"] == 2:
                buf = status[(o["offset"] // 4):(o["offset"] // 4) + 1]
            elif o["phase"] == 0:
                first = o["offset"] // 8
                buf = ins[c][first:first + cnt] if root else st_in[c][(k & 1) * slot_in:(k & 1) * slot_in + cnt]
                if not root:
                    after.append((c, k, first, cnt))
            else:
                first = o["offset"] // 8
                buf = outs[c][first:first + cnt] if root else st_out[c][(k & 1) * slot_out:(k & 1) * slot_out + cnt]
            ops.append(dist.P2POp(dist.isend if o["kind"] == 0 else dist.irecv, buf, o["peer"]))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        for c, k, first, cnt in after:   # the owner's stand-in filter of chunk k: samples on the decimation grid
            x = st_in[c][(k & 1) * slot_in:(k & 1) * slot_in + cnt]
            skip = (decim - (consumed + first) % decim) % decim
            y = x[skip::decim]
            st_out[c][(k & 1) * slot_out:(k & 1) * slot_out + y.numel()] = y
    dist.barrier()
    ok = None
    if root:
        ok = True
        for c in range(channels):
            if c % world == 0:
                continue
            ok = ok and bool(torch.equal(outs[c], pattern(c)[n0::decim]))
        ok = ok and bool((status == 0).all())
    # the record of the real leg, filled from this rehearsal: per-rank times of the walk above, bytes from the plan itself
    t_all = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(t_all, torch.tensor([(time.perf_counter() - t_walk) * 1e3], dtype=torch.float64))
    identical = None
    if root:
        identical = [True if c % world == 0 else bool(torch.equal(outs[c], pattern(c)[n0::decim])) for c in range(channels)]
    check = multi_gpu_check_record(world, channels, samples, taps_n, decim, "gloo (dry run: the plan's operations as batch_isend_irecv)",
                                   [[float(t[0]) for t in t_all]], identical, outputs=m_total)
    sent = sum(o["bytes"] for o in plan if o["kind"] == 0)
    recv = sum(o["bytes"] for o in plan if o["kind"] == 1)
    # (the plan of THIS rank must move exactly the bytes the record's formula states for it)
    plan_agrees = sent == check["bytes_sent_per_rank"][rank] and recv == check["bytes_received_per_rank"][rank]
    agree = torch.tensor([1 if plan_agrees else 0], dtype=torch.int32)
    dist.all_reduce(agree, op=dist.ReduceOp.MIN)
    return {"dry_run": True, "world": world, "channels": channels, "chunks": len({o["chunk"] for o in plan if o["phase"] == 0}),
            "groups": int(gmax.item()) + 1, "ok": ok, "plan_bytes_agree_with_record": bool(agree.item()),
            "multi_gpu_check": check,
            "note": "if_fir_mc_debug_plan executed group by group over gloo with two-slot staging and a stand-in filter"}


def committed_traffic(workload, backend_name):
    """profiles/traffic.json record of a workload's kernel (TCC counters of committed rocprofv3 passes), or None."""
    try:
        key = workload + (":fir_fft" if backend_name == "hip_fft" else ":fir_direct")
        return json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(key)
    except Exception:   # noqa: BLE001
        return None


def time_config(fir, name, backend, x, dev, stream, steps, warmup, names):
    """One more BASELINE config timed on the resident synthetic stream (extra.configs; never part of `value`)."""
    taps_n, decim, log2n, desc = WORKLOADS[name]
    n = 1 << log2n
    taps = fir.bpf_design(taps_n)
    with fir.IfFir(taps, decim, 0, device=dev.index, backend=backend) as fc:
        fc.set_stream(stream.cuda_stream)
        y = torch.empty(2 * fc.out_count(n), dtype=torch.float32, device=dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(warmup):
            fc.process_device(x.data_ptr(), y.data_ptr(), n)
        e1.record(stream)
        torch.cuda.synchronize()
        # a kernel's rate settles only after ~100 ms of ITS OWN launches (power management): keep going for that long
        # when a launch is short (the 34 ms tap-split launches are not repeated for this)
        per = e0.elapsed_time(e1) / max(warmup, 1)
        if per < 5.0:
            for _ in range(int(min(400, 100.0 / max(per, 0.05)))):
                fc.process_device(x.data_ptr(), y.data_ptr(), n)
        e0.record(stream)
        for _ in range(steps):
            fc.process_device(x.data_ptr(), y.data_ptr(), n)
        e1.record(stream)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / steps
        nonzero = bool((y[:1 << 16] != 0).any().item())
        b = names[fc.get_backend()]
    del y
    gbs = algorithmic_bytes_per_sample(decim) * n / (ms * 1e-3) / 1e9
    return {"backend": b, "kernel_ms": round(ms, 4), "steps": steps, "msamples_per_s": round(n / ms / 1e3, 1),
            "hbm_gbs": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4),
            "valu_tflops_direct_form_equivalent": round(algorithmic_flops_per_sample(taps_n, decim) * n / (ms * 1e-3) / 1e12, 2),
            "output_nonzero": nonzero}


def time_filter_bank(fir, x, dev, stream, channels=8, taps_n=255, decim=8, log2n=28, steps=20, own_centres=False):
    """The filter bank (if_fir_channelizer_process_device: `channels` fs/16 channels from ONE pass over the resident wideband
    stream, decimation 8 = 2x oversampled) timed like an extra config, channel 0 checked against a context that mixes, filters
    and decimates that one channel (extra.filter_bank; never part of `value`).  own_centres (round 5, extra.filter_bank_own_centres):
    every channel at its own centre on the fs/4096 grid, if_fir_channelizer_process_device_freq -- what QO-100's narrow-band
    channels need (they do not sit on fs/16)."""
    n = min(1 << log2n, x.numel() // 2)
    taps = fir.bpf_design(taps_n, 0.0, 0.02)
    slots = [(2 * c + 1) % 16 for c in range(channels)] if channels <= 8 else list(range(channels))
    centres = [(((256 * s + 37 + 11 * c) + 2048) % 4096 - 2048) / 4096.0 for c, s in enumerate(slots)]

    def bank(fb, ptrs):
        if own_centres:
            return fb.channelizer_process_device_freq(centres, x.data_ptr(), ptrs, n)
        return fb.channelizer_process_device(slots, x.data_ptr(), ptrs, n)
    with fir.IfFir(taps, decim, 0, device=dev.index) as fb:
        fb.set_stream(stream.cuda_stream)
        m = fb.out_count(n)
        outs = [torch.empty(2 * m, dtype=torch.float32, device=dev) for _ in range(channels)]
        ptrs = [o.data_ptr() for o in outs]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(100):   # ~100 ms of its own launches
            bank(fb, ptrs)
        e0.record(stream)
        for _ in range(steps):
            bank(fb, ptrs)
        e1.record(stream)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / steps
        fb.reset()
        bank(fb, ptrs)
        fb.synchronize()
    with fir.IfFir(taps, decim, 0, device=dev.index) as f1:
        s0 = slots[0]
        f1.set_nco(centres[0] if own_centres else (s0 / 16.0 if s0 <= 8 else s0 / 16.0 - 1.0))
        f1.set_stream(stream.cuda_stream)
        ref = torch.empty(2 * m, dtype=torch.float32, device=dev)
        f1.process_device(x.data_ptr(), ref.data_ptr(), n)
        f1.synchronize()
        rel = ((ref - outs[0]).abs().max() / ref.abs().max()).item()
    del outs, ref
    bytes_alg = (8.0 + channels * 8.0 / decim) * n
    return {"workload": "%d channels (%s) x (%d-tap prototype, decimate-by-%d) from one 2^%d-sample stream" %
                        (channels, "centres %s on the fs/4096 grid" % [round(c, 5) for c in centres] if own_centres else "slots %s" % slots,
                         taps_n, decim, log2n),
            "kernel_ms": round(ms, 4), "steps": steps, "input_msamples_per_s": round(n / ms / 1e3, 1),
            "algorithmic_bytes": bytes_alg, "hbm_gbs": round(bytes_alg / (ms * 1e-3) / 1e9, 1),
            "frac": round(bytes_alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "channel0_max_rel_diff_vs_nco_context": rel, "ok": bool(rel <= 2e-6)}


def main():
    # Everything except the final JSON line goes to stderr: RCCL prints a version banner on stdout at communicator
    # creation, which would break the one-line contract.
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="fir255_dec4_2p28", choices=sorted(WORKLOADS))
    ap.add_argument("--variant", type=int, default=None, help="kernel tuning variant (if_fir_set_tuning)")
    ap.add_argument("--backend", default="auto", choices=["auto", "direct", "fft", "generic"],
                    help="kernel family (auto = library default: overlap-save FFT where it applies)")
    ap.add_argument("--scatter", action="store_true", help="also time RCCL scatter/gather of channels from rank 0")
    ap.add_argument("--scatter-mc", action="store_true",
                    help="like --scatter through the library's own multi-channel front (if_fir_mc_*, in-library RCCL); "
                         "reported in \"extra\", never in `value`")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--channels", type=int, default=None,
                    help="fixed TOTAL number of channels, channel c on rank c mod N, each rank filters its channels back "
                         "to back (BASELINE configs[3]: 8 channels on 8/4/2/1 GPUs; strong scaling).  Default: one "
                         "channel per GPU (weak scaling)")
    ap.add_argument("--dry-run", action="store_true",
                    help="rehearse the launch path on the CPU (gloo, a sleep instead of the filter); no GPU needed")
    ap.add_argument("--condition-ms", type=float, default=300.0,
                    help="device time of untimed GPU work ahead of the warm-up steps (settled power state; see main)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not measure roofline.traffic in this run (two short rocprofv3 --pmc child passes ahead of the "
                         "benchmark); the committed counter passes of profiles/traffic.json are quoted instead")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="skip the other single-GPU BASELINE configs timed after the headline (extra.configs)")
    ap.add_argument("--extras-first", action="store_true",
                    help="development: run the comparison measurements (extra.*) between the two halves of the conditioning, "
                         "ahead of the timed steps, as up to round 3 (A/B of the order: profiles/r04_bench_order.txt)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # not under a launcher: become the launcher (child process; nothing here has touched the GPU yet)
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    if args.dry_run:
        return dry_run(args, rank, world, json_fd)
    if args.traffic_child:
        return traffic_child(args)
    # roofline.traffic measured on THIS box in THIS run: profiler passes of a short child run, before
    # this process touches the GPU.  One GPU only (the children use device 0); any failure falls back to the committed passes.
    live = None
    if world == 1 and not args.no_live_traffic and not args.channels and args.variant is None:
        try:
            live = live_traffic(args.workload, args.backend)
        except Exception:   # noqa: BLE001 - never fatal
            live = None
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: libif_fir has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # IF_FIR_BENCH_DIST=1 runs the collective code path (RCCL init, barriers, MAX-reduce) even with one rank, so it can
    # be rehearsed on a one-GPU box
    use_dist = world > 1 or os.environ.get("IF_FIR_BENCH_DIST") == "1"
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        # the first collective of a process sets the transport up (6.5 ms on one MI355X, tools/barrier_probe.py; later ones
        # take 30 us): spent here, not in front of the timed steps, where an idle device falls out of its settled power state
        dist.barrier()
        dist.barrier()

    pkg = graft.load_pkg()
    fir = pkg.if_fir
    taps_n, decim, log2n, desc = WORKLOADS[args.workload]
    n = 1 << log2n
    i16 = "_i16_" in args.workload
    # NCO workload: a frequency whose phase word is a multiple of 2^19, so that P * 8192 = 0 mod 2^32 and the
    # whole-output check can reproduce the timed context's phase from a short priming call
    nco = 1638.0 / 8192.0 if "_nco_" in args.workload else 0.0
    in_bytes = 4.0 if i16 else 8.0
    taps = fir.bpf_design(taps_n)
    backend_ids = {"auto": fir.BACKEND_AUTO, "direct": fir.BACKEND_HIP_DIRECT, "fft": fir.BACKEND_HIP_FFT,
                   "generic": fir.BACKEND_HIP_GENERIC}
    f = fir.IfFir(taps, decim, 0, device=local_rank, backend=backend_ids[args.backend])
    if args.variant is not None:
        f.set_tuning(args.variant)
    if nco:
        f.set_nco(nco)
    stream = torch.cuda.Stream(device=dev)   # a real (non-null) HIP stream: handle 0 would mean "context's own stream"
    torch.cuda.set_stream(stream)
    f.set_stream(stream.cuda_stream)         # kernels run on this stream so the torch.cuda.Event pair brackets them
    m = f.out_count(n)
    x = torch.empty(2 * n, dtype=torch.float32, device=dev)
    y = torch.empty(2 * m, dtype=torch.float32, device=dev)
    total_channels = args.channels if args.channels else world
    if total_channels < world:
        raise SystemExit("--channels must be at least the number of GPUs")
    owned = [c for c in range(total_channels) if c % world == rank]
    channel = owned[0]                  # default: one independent transponder channel per GPU (weak scaling)
    f.synth_device(x.data_ptr(), 0, n, channel)
    torch.cuda.synchronize()
    more = []                           # further channels of this rank (--channels): own context, input and output
    for c in owned[1:]:
        if i16 or nco:
            raise SystemExit("--channels is implemented for the float32, NCO-less workloads")
        fc = fir.IfFir(taps, decim, 0, device=local_rank, backend=backend_ids[args.backend])
        if args.variant is not None:
            fc.set_tuning(args.variant)
        fc.set_stream(stream.cuda_stream)
        xc = torch.empty(2 * n, dtype=torch.float32, device=dev)
        fc.synth_device(xc.data_ptr(), 0, n, c)
        more.append((fc, xc, torch.empty(2 * m, dtype=torch.float32, device=dev)))
    torch.cuda.synchronize()
    if i16:
        # int16 front-end: quantise the synthetic stream (full scale = 2.0) on the device, filter the int16 buffer
        f.set_input_format(fir.INPUT_I16)
        xq = torch.empty(2 * n, dtype=torch.int16, device=dev)
        step_q = 1 << 24
        for a in range(0, 2 * n, step_q):
            xq[a:a + step_q] = torch.clamp(torch.round(x[a:a + step_q] * 16384.0), -32768, 32767).to(torch.int16)
        x_f32_head = x[:2 * (1 << 16)].clone()
        del x
        x = xq
        torch.cuda.synchronize()

    # every step continues the stream (history + phase carried): identical work per step, no reset memset
    def step_stream():
        f.process_device(x.data_ptr(), y.data_ptr(), n)
        for fc, xc, yc in more:
            fc.process_device(xc.data_ptr(), yc.data_ptr(), n)

    # ---- the COLD figure: the driver's own W warm-up + K steps right away, before any conditioning ------
    # Reported beside `value` (roofline.cold_*), never as `value`: what a caller sees who starts filtering on an idle chip.
    cold_ms = None
    if args.condition_ms > 0:
        for _ in range(args.warmup):
            step_stream()
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record(stream)
        for _ in range(args.steps):
            step_stream()
        c1.record(stream)
        torch.cuda.synchronize()
        cold_ms = c0.elapsed_time(c1) / args.steps
    cold_passes = (args.warmup + args.steps) if cold_ms is not None else 0

    # ---- device conditioning BEFORE the headline ----------------------------------------------------------------------
    # After idling, the chip's power management first boosts, then clamps the clock hard for ~20 launches (10 ms) and
    # only then settles (profiles/r02a_kernel_trace_summary.json: 0.48-0.52 ms, then 0.60-0.67 ms, then 0.49 ms per
    # launch).  A filter in service streams continuously, so `value` is the settled rate: every rank first keeps its
    # GPU busy for >= --condition-ms of device time with untimed passes of the very same step.  The W warm-up steps and the K
    # timed steps follow unchanged; the other things this script measures (direct form, the other BASELINE configs, a plain
    # copy, the filter bank: "extra", never part of `value`) come after them (measure_extras; --extras-first: the order up to round 3).
    extra = {}
    names = {1: "hip_direct", 2: "hip_tapsplit", 3: "hip_generic", 4: "hip_fft"}
    cond0 = torch.cuda.Event(enable_timing=True)
    cond0.record(stream)
    cond_passes = 0
    cond1 = torch.cuda.Event(enable_timing=True)

    def condition(until_ms, min_passes):
        # untimed passes of the very same step until `until_ms` of device time have passed since cond0
        nonlocal cond_passes
        done = 0
        while args.condition_ms > 0 and cond_passes < 100000:
            cond1.record(stream)
            torch.cuda.synchronize()
            if cond0.elapsed_time(cond1) >= until_ms and done >= min_passes:
                break
            for _ in range(20):
                step_stream()
            cond_passes += 20
            done += 20

    condition(args.condition_ms / 2, 20)
    def measure_extras():
        # The comparison measurements (direct form, the other BASELINE configs, a plain copy, the filter bank: reported under
        # "extra", never part of `value`).  Round 4: they run AFTER the timed steps.  They used to sit between the two halves of the
        # conditioning; every second of full-power work ahead of the timed region costs the headline (same box, same library,
        # driver form: 0.4806 ms with four comparison measurements ahead of it, 0.5103 ms with a fifth -- the chip is at its
        # package power limit and warmer silicon leaks more; profiles/r04_bench_order.txt).  Each of them settles on ~100 ms of
        # its own launches.
        if args.backend == "auto" and f.get_backend() == fir.BACKEND_HIP_FFT and taps_n in (127, 255) \
                and decim in (1, 4) and not i16 and not nco:
            # the north_star's direct-form MAC kernel, timed beside the default overlap-save path (same buffers, same
            # stream; not part of `value`)
            f.set_backend(fir.BACKEND_HIP_DIRECT)
            f.reset()
            for _ in range(max(args.warmup, 100)):   # ~100 ms of its own launches: settled like the headline
                step_stream()
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(args.steps):
                step_stream()
            e1.record(stream)
            torch.cuda.synchronize()
            dms = e0.elapsed_time(e1) / args.steps
            extra["direct_form"] = {
                "backend": "hip_direct", "kernel_ms": round(dms, 4), "msamples_per_s": round(n / dms / 1e3, 1),
                "hbm_frac": round(algorithmic_bytes_per_sample(decim) * n / (dms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "valu_tflops": round(algorithmic_flops_per_sample(taps_n, decim) * n / (dms * 1e-3) / 1e12, 2),
                "note": "hand-written v_pk_fma_f32 direct form (bit-exact vs the oracle's float32 order model); power-limited"}
            f.set_backend(fir.BACKEND_AUTO)
            f.reset()
        if world == 1 and not i16:
            # what this very box's HBM does on plain streaming kernels (context for roofline.frac: the nominal peak is
            # 8 TB/s, a device-to-device copy of the same buffer reaches about two thirds of it)
            try:
                xs = x[:min(x.numel(), 1 << 29)]
                dst = torch.empty_like(xs)
                cs0, cs1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for _ in range(3):
                    dst.copy_(xs)
                cs0.record()
                for _ in range(10):
                    dst.copy_(xs)
                cs1.record()
                torch.cuda.synchronize()
                copy_ms = cs0.elapsed_time(cs1) / 10
                extra["hbm_copy_on_this_box"] = {"gbs": round(2 * xs.numel() * 4 / (copy_ms * 1e-3) / 1e9, 1),
                                                 "frac_of_peak": round(2 * xs.numel() * 4 / (copy_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                                 "note": "torch device-to-device copy of %d MiB (read + write bytes / time)" % (xs.numel() * 4 >> 20)}
                del dst
            except Exception as e:   # noqa: BLE001 - context only, never fatal
                extra["hbm_copy_on_this_box"] = {"error": repr(e)}
        if world == 1 and not args.no_extra_configs and args.workload == "fir255_dec4_2p28" and not args.channels:
            # the other single-GPU BASELINE configs on the same resident stream: AUTO and the comparison form
            # (direct form where the unrolled kernels exist, else the tap-split kernel of the north_star's wording)
            cfgs = {}
            try:
                torch.cuda.empty_cache()
                for cname, forms in (("fir127_2p26", (("auto", fir.BACKEND_AUTO, 20, 5), ("direct", fir.BACKEND_HIP_DIRECT, 20, 5))),
                                     ("fir1023_2p28", (("auto", fir.BACKEND_AUTO, 20, 5), ("tapsplit", fir.BACKEND_HIP_TAPSPLIT, 2, 1))),
                                     # (round 4: an odd decimation on its own kernel -- blocks of 3 x 1024 samples -- beside the BASELINE configs)
                                     ("fir255_dec3_2p28", (("auto", fir.BACKEND_AUTO, 20, 5),))):
                    cfgs[cname] = {"workload": WORKLOADS[cname][3]}
                    for label, b, st, wu in forms:
                        cfgs[cname][label] = time_config(fir, cname, b, x, dev, stream, st, wu, names)
                    # HBM bytes per launch of this config's AUTO kernel from the committed counter passes (another box)
                    rec = committed_traffic(cname, cfgs[cname]["auto"]["backend"])
                    if rec:
                        cfgs[cname]["auto"]["traffic"] = rec.get("hbm_bytes_per_launch")
                        cfgs[cname]["auto"]["traffic_over_algorithmic"] = round(
                            rec.get("hbm_bytes_per_launch") / (algorithmic_bytes_per_sample(WORKLOADS[cname][1]) * (1 << WORKLOADS[cname][2])), 4)
                        cfgs[cname]["auto"]["traffic_source"] = "profiles/traffic.json (%s, rocprofv3 FETCH_SIZE/WRITE_SIZE passes)" % rec.get("round")
            except Exception as e:   # noqa: BLE001 - context only, never fatal
                cfgs["error"] = repr(e)
            extra["configs"] = cfgs
            # 8 channels from one pass over the same resident stream (round 4: the all-slots form)
            try:
                torch.cuda.empty_cache()
                if i16:
                    raise RuntimeError("float32 stream only")
                extra["filter_bank"] = time_filter_bank(fir, x, dev, stream, log2n=WORKLOADS[args.workload][2])
            except Exception as e:   # noqa: BLE001 - context only, never fatal
                extra["filter_bank"] = {"error": repr(e)}
            # round 5: the same eight channels at their own centres (general forms), decimation 8, 16 and 64
            fbo = {}
            for dec_o in (8, 16, 64):
                try:
                    torch.cuda.empty_cache()
                    if i16:
                        raise RuntimeError("float32 stream only")
                    fbo["decimate_by_%d" % dec_o] = time_filter_bank(fir, x, dev, stream, decim=dec_o, log2n=WORKLOADS[args.workload][2],
                                                                     own_centres=True)
                except Exception as e:   # noqa: BLE001 - context only, never fatal
                    fbo["decimate_by_%d" % dec_o] = {"error": repr(e)}
            extra["filter_bank_own_centres"] = fbo
    if args.extras_first:
        measure_extras()
    condition(args.condition_ms, 20)       # and passes of the step itself right in front of the warm-up
    cond1.record(stream)
    torch.cuda.synchronize()
    conditioning_ms = cond0.elapsed_time(cond1)

    for _ in range(args.warmup):
        step_stream()
    if use_dist:
        # the opening barrier's host round trip runs while the device is still busy with (untimed) passes of the step:
        # an idle gap here costs the settled power state the conditioning created, and a multi-rank run would then time
        # "cold" kernels where the one-rank run times settled ones (measured with IF_FIR_BENCH_DIST=1 on one GPU:
        # 0.583 instead of 0.476 ms per step)
        for _ in range(BARRIER_COVER_STEPS):
            step_stream()
        dist.barrier()
    torch.cuda.synchronize()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    step_ev = []
    for _ in range(args.steps):
        step_stream()
        step_ev.append(torch.cuda.Event(enable_timing=True))
        step_ev[-1].record(stream)
    ev1.record(stream)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    dev_ms_per_step = ev0.elapsed_time(ev1) / args.steps
    per_step = [a.elapsed_time(b) for a, b in zip([ev0] + step_ev[:-1], step_ev)]   # this rank's steps, HIP events
    t = torch.tensor([wall, dev_ms_per_step, cold_ms if cold_ms is not None else 0.0], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall_max, dev_ms_max, cold_ms_max = float(t[0].item()), float(t[1].item()), float(t[2].item())

    # whole-output check of the TIMED context's last step against a different kernel family in the same stream state
    # (a work-distribution bug that leaves blocks unwritten makes a launch look fast; windows do not see it)
    whole = whole_output_check(fir, f, taps, decim, x, y, n, i16, stream, local_rank,
                               cold_passes + cond_passes + args.warmup + args.steps, names, nco)
    if use_dist:
        okt = torch.tensor([1.0 if whole["ok"] else 0.0], dtype=torch.float64, device=dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        whole["ok_all_ranks"] = bool(okt.item() > 0.5)
    if not whole.get("ok_all_ranks", whole["ok"]):
        if rank == 0:
            os.write(json_fd, (json.dumps({"metric": "complex-IQ MSamples/s through %d-tap FIR" % taps_n, "value": None,
                                           "error": "timed output failed the whole-output parity check",
                                           "parity": {"whole_output": whole}}) + "\n").encode())
        f.close()
        if use_dist:
            dist.destroy_process_group()
        sys.exit(1)
    if not args.extras_first:
        measure_extras()
    if args.scatter and use_dist:
        cs = pkg.channel_shard
        root_inputs = None
        if rank == 0:
            root_inputs = [x] + [torch.empty_like(x) for _ in range(world - 1)]
            for c in range(1, world):
                f.synth_device(root_inputs[c].data_ptr(), 0, n, c)
            torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        ts = time.perf_counter()
        mine = cs.scatter_channels(root_inputs, world, n, dev, root=0)
        torch.cuda.synchronize()
        dist.barrier()
        t_scatter = time.perf_counter() - ts
        outs = {}
        for c, xin in mine.items():
            yo = torch.empty(2 * f.out_count(n), dtype=torch.float32, device=dev)
            f.process_device(xin.data_ptr(), yo.data_ptr(), n)
            outs[c] = yo
        torch.cuda.synchronize()
        dist.barrier()
        t_filter = time.perf_counter() - ts - t_scatter
        cs.gather_outputs(outs, world, m, dev, root=0)
        torch.cuda.synchronize()
        dist.barrier()
        t_all = time.perf_counter() - ts
        extra["scatter_gather"] = {"scatter_s": t_scatter, "filter_s": t_filter, "gather_s": t_all - t_scatter - t_filter,
                                   "end_to_end_msamples_per_s": world * n / t_all / 1e6,
                                   "note": "RCCL grouped send/recv of whole channels from rank 0 over xGMI"}

    if args.scatter_mc and use_dist and not (i16 or nco):
        # the C front does the same movement itself: bootstrap id over torch.distributed, then one collective call
        idt = torch.zeros(fir.MC_ID_BYTES, dtype=torch.uint8, device=dev)
        if rank == 0:
            idt = torch.frombuffer(bytearray(fir.mc_unique_id()), dtype=torch.uint8).to(dev)
        dist.broadcast(idt, 0)
        # BASELINE configs[3] in one call: all `total_channels` channels resident on rank 0, channel c filtered by rank c mod world
        mc_channels = total_channels
        taps_all = np.stack([taps] * mc_channels)
        with fir.IfFirMc(taps_all, decim, n, device=local_rank, rank=rank, world=world,
                         unique_id=bytes(idt.cpu().numpy().tobytes())) as mc:
            ins = outs_mc = None
            if rank == 0:
                ins = [x] + [torch.empty_like(x) for _ in range(mc_channels - 1)]
                for c in range(1, mc_channels):
                    f.synth_device(ins[c].data_ptr(), 0, n, c)
                outs_mc = [torch.empty(2 * m, dtype=torch.float32, device=dev) for _ in range(mc_channels)]
                torch.cuda.synchronize()
            times, per_rank = [], []
            for _ in range(3):
                mc.reset()
                dist.barrier()
                ts = time.perf_counter()
                mc.process_device([t_.data_ptr() for t_ in ins] if rank == 0 else None,
                                  [t_.data_ptr() for t_ in outs_mc] if rank == 0 else None, n)
                mine_ms = (time.perf_counter() - ts) * 1e3
                dist.barrier()
                times.append(time.perf_counter() - ts)
                t_all = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
                dist.all_gather(t_all, torch.tensor([mine_ms], dtype=torch.float64, device=dev))
                per_rank.append([float(t_[0]) for t_ in t_all])
            identical = None
            if rank == 0:
                # every channel against a single-channel context fed the same (first) call
                identical = []
                for c in range(mc_channels):
                    f.reset()
                    f.process_device(ins[c].data_ptr(), y.data_ptr(), n)
                    torch.cuda.synchronize()
                    identical.append(bool(torch.equal(y, outs_mc[c])))
                f.reset()
            extra["scatter_gather_mc"] = {"end_to_end_s": min(times), "end_to_end_msamples_per_s": mc_channels * n / min(times) / 1e6,
                                          "channel0_identical_to_single_context": identical[0] if identical else None,
                                          "multi_gpu_check": multi_gpu_check_record(world, mc_channels, n, taps_n, decim,
                                                                                    "rccl (ncclSend/ncclRecv inside libif_fir.so)",
                                                                                    per_rank, identical),
                                          "note": "if_fir_mc_process_device: grouped RCCL send/recv from rank 0 inside libif_fir.so"}

    if rank == 0:
        ms_per_step = wall_max / args.steps * 1e3
        value = total_channels * n / (wall_max / args.steps) / 1e6
        # per step on rank 0: its channels back to back (one kernel each)
        bytes_per_launch = algorithmic_bytes_per_sample(decim, in_bytes) * n * len(owned)
        flops_per_launch = algorithmic_flops_per_sample(taps_n, decim) * n * len(owned)
        achieved_gbs = bytes_per_launch / (dev_ms_max * 1e-3) / 1e9
        achieved_tf = flops_per_launch / (dev_ms_max * 1e-3) / 1e12
        traffic = traffic_src = None
        if live is not None:
            traffic = live["hbm_bytes_per_launch"]
            traffic_src = ("measured in this run on this box: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes (one counter "
                           "per pass) of a 3-step child run ahead of the benchmark, kernel %s, %d + %d launches; read bytes = "
                           "2 x FETCH_SIZE (gfx950 note of MI355X_MICROARCH.md): %.4g GB read + %.4g GB written"
                           % (live["kernel"], live["launches_averaged"][0], live["launches_averaged"][1],
                              live["read_bytes_per_launch"] / 1e9, live["write_bytes_per_launch"] / 1e9))
        else:
            rec = committed_traffic(args.workload, names[f.get_backend()])
            if rec and rec.get("hbm_bytes_per_launch") is not None:
                traffic = rec.get("hbm_bytes_per_launch") * len(owned)   # a step launches one kernel per owned channel
                traffic_src = ("NOT measured in this run: TCC counters of kernel %s from the committed rocprofv3 "
                               "passes profiles/%s_* (another box), file profiles/traffic.json"
                               % (rec.get("kernel"), rec.get("round")))
        # parity spot check (oracle as checker only): first 4096 outputs of this rank's last step vs the order model.
        # the stream was continued for warmup+steps calls, so regenerate the expected state cheaply: only check that
        # a fresh context reproduces the oracle on the head of the stream.
        oracle = graft.load_oracle()
        f2 = fir.IfFir(taps, decim, 0, device=local_rank, backend=backend_ids[args.backend])
        f2.set_stream(stream.cuda_stream)
        if i16:
            f2.set_input_format(fir.INPUT_I16)
        if args.variant is not None:
            f2.set_tuning(args.variant)
        if nco:
            f2.set_nco(nco)
        head = 1 << 16
        yh = torch.empty(2 * f2.out_count(head), dtype=torch.float32, device=dev)
        f2.process_device(x.data_ptr(), yh.data_ptr(), head)
        torch.cuda.synchronize()
        xh = x[:2 * head].cpu().numpy()
        if i16:
            xh = xh.astype(np.float32) * np.float32(2.0 ** -15)
        ref64 = oracle.fir_nco_f64(taps, xh, decim, oracle.nco_phase_word(nco)) if nco else oracle.fir_f64(taps, xh, decim)
        l2, mx = oracle.err_metrics(yh.cpu().numpy(), ref64)
        parity = {"rel_l2_vs_f64_oracle": l2, "rel_max_vs_f64_oracle": mx, "tolerance": 1e-6,
                  "window": "first 2^16 input samples", "whole_output": whole}
        if f2.get_backend() in (fir.BACKEND_HIP_DIRECT, fir.BACKEND_HIP_GENERIC) and not nco:
            model = oracle.fir_f32fma(taps, xh, decim, seg_mode=1, seg_len=32)
            parity["bit_exact_vs_f32_order_model"] = bool(np.array_equal(yh.cpu().numpy(), model))
        f2.close()
        line = {
            "metric": "complex-IQ MSamples/s through %d-tap FIR" % taps_n,
            "value": round(value, 1), "unit": "MSamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "cold_ms_per_step": round(cold_ms_max, 4) if cold_ms is not None else None,
            "higher_is_better": True, "scaling": "strong" if args.channels else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": desc, "name": args.workload, "taps": taps_n, "decimation": decim,
                       "samples_per_channel": n, "channels": total_channels, "channels_on_rank0": len(owned),
                       "parallelism": "channel c on rank c mod %d, no data-path collective" % world,
                       "backend": names[f.get_backend()],
                       "device": f.device_info()},
            "conditioning": {"device_ms": round(conditioning_ms, 1), "extra_passes_of_the_step": cond_passes,
                             "passes_covering_the_opening_barrier": BARRIER_COVER_STEPS if use_dist else 0,
                             "extras_order": "before" if args.extras_first else "after",
                             "note": "untimed GPU work ahead of the W warm-up steps: passes of the same step (and, only with "
                                     "--extras-first = extras_order 'before', the comparison measurements, as in the records up to "
                                     "round 3; since round 4 they run BEHIND the timed steps, which alone moves `value` by about "
                                     "6 %: profiles/r04_bench_order.txt) so that the timed steps run in the settled power state; "
                                     "multi-rank runs queue further untimed passes in front of the opening barrier so "
                                     "that the device is not idle during its host round trip"},
            "roofline": {"bound": "hbm", "achieved": round(achieved_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved_gbs / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_over_algorithmic": round(traffic / bytes_per_launch, 4) if traffic else None,
                         "traffic_source": traffic_src,
                         "kernel_ms": round(dev_ms_max, 4),
                         "cold_kernel_ms": round(cold_ms_max, 4) if cold_ms is not None else None,
                         "cold_frac": round(bytes_per_launch / (cold_ms_max * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if cold_ms is not None else None,
                         "cold_note": "the same W warm-up + K steps timed BEFORE any conditioning, i.e. launches ~6-25 after idle: "
                                      "the power controller's transient (launch by launch, profiles/r03_cold_transient.txt: "
                                      "launches 2-3 after idle 0.44 ms, launches 5-10 0.67-0.72 ms, sustained 0.47-0.48 ms "
                                      "again after ~35 ms); `value` and `frac` are the sustained rate of a filter in "
                                      "continuous service",
                         "kernel_ms_median": round(float(np.median(per_step)), 4),
                         "kernel_ms_min": round(float(np.min(per_step)), 4),
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "note": "HIP events on the launch stream around the timed steps / steps (everything a step "
                                 "launches: one kernel on the overlap-save path)"},
            "valu": {"achieved": round(achieved_tf, 2), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(achieved_tf / VALU_PEAK_TFLOPS, 4),
                     "note": ("direct-form-EQUIVALENT rate (4*taps/decimation flop per input sample / time); the "
                              "overlap-save kernel executes ~8x fewer flops, so this is not a VALU utilisation")
                     if f.get_backend() == fir.BACKEND_HIP_FFT else "executed FP32 VALU flops / time"},
            "parity": parity,
        }
        if extra:
            line["extra"] = extra
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(taps, decim)
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    f.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
