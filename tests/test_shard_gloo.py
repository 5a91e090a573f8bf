"""world_size-2 gloo test of the channel-parallel host logic: channel map, grouped point-to-point
scatter of whole channels from the root, per-rank filtering, gather.  On CPU the per-channel filter is the oracle
(test stand-in for IfFir.process_device, which needs a GPU)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_channels, samples, decim, q):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    pkg = g.load_pkg()
    oracle = g.load_oracle()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cs = pkg.channel_shard
        taps = oracle.bpf_design(127)
        dev = torch.device("cpu")
        root_inputs = None
        if rank == 0:
            root_inputs = [torch.from_numpy(oracle.synth_iq(samples, channel=c)) for c in range(n_channels)]
        mine = cs.scatter_channels(root_inputs, n_channels, samples, dev, root=0)
        assert sorted(mine) == cs.channel_map(n_channels, world)[rank]
        for c, t in mine.items():   # every rank received exactly its channels' streams
            assert np.array_equal(t.numpy(), oracle.synth_iq(samples, channel=c))

        def filt(c, x):
            return torch.from_numpy(oracle.fir_f32fma(taps, x.numpy(), decim, seg_mode=1, seg_len=32))

        outs = cs.filter_channels(mine, filt)
        m = oracle.out_count(0, samples, decim)
        res = cs.gather_outputs(outs, n_channels, m, dev, root=0)
        if rank == 0:
            ok = all(np.array_equal(res[c].numpy(),
                                    oracle.fir_f32fma(taps, oracle.synth_iq(samples, channel=c), decim, seg_mode=1,
                                                      seg_len=32)) for c in range(n_channels))
            q.put(ok)
        else:
            assert res is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_channels,decim", [(8, 4), (3, 1), (1, 4)])
def test_channel_scatter_filter_gather_world2(n_channels, decim):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_channels, 2048, decim, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_channel_map():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    cs = g.load_pkg().channel_shard
    assert cs.channel_map(8, 8) == [[c] for c in range(8)]
    assert cs.channel_map(8, 2) == [[0, 2, 4, 6], [1, 3, 5, 7]]
    assert cs.channel_map(3, 4) == [[0], [1], [2], []]
    assert cs.channel_map(0, 2) == [[], []]
    assert sorted(sum(cs.channel_map(13, 4), [])) == list(range(13))
    with pytest.raises(ValueError):
        cs.channel_map(1, 0)
