"""Channel-parallel sharding of the FIR path over the GPUs of one node (BUILD-DEFINED).

Each transponder channel is an independent (input stream, taps) pair, so the path shards by channel with no
data-path reduction: channel c runs on rank c mod world.  One process per GPU (torch.distributed; backend "nccl"
is RCCL over xGMI on ROCm, "gloo" in the CPU tests).  A collective is issued only when more than one channel is
filtered AND the inputs live on one root rank: the root scatters whole channels with one grouped batch of
point-to-point sends (RCCL groups them into one ncclGroupStart/End, so the root drives its 7 xGMI links
concurrently — xGMI is point-to-point, there is no switch to broadcast through), and outputs are gathered the
same way.  When every rank produces its own input (device-resident synthetic data, or one SDR per GPU) there is
no collective at all.

The reference has no counterpart (no collective, no multi-process code: SURVEY.md §2 "Native-code, CUDA and
collective inventory").
"""
from typing import Callable, Dict, List, Optional, Sequence

import torch
import torch.distributed as dist


def channel_map(n_channels: int, world_size: int) -> List[List[int]]:
    """channels owned by each rank: c -> rank c mod world."""
    if n_channels < 0 or world_size < 1:
        raise ValueError("bad channel_map arguments")
    return [[c for c in range(n_channels) if c % world_size == r] for r in range(world_size)]


def owner_of(channel: int, world_size: int) -> int:
    return channel % world_size


def scatter_channels(root_inputs: Optional[Sequence[torch.Tensor]], n_channels: int, samples: int,
                     device: torch.device, root: int = 0, group=None) -> Dict[int, torch.Tensor]:
    """Root holds `n_channels` interleaved-IQ float32 tensors of 2*samples floats; every rank returns
    {channel: tensor} for the channels it owns.  No-op (no collective) when world == 1 or n_channels == 1 and
    root owns it."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    mine = channel_map(n_channels, world)[rank]
    out: Dict[int, torch.Tensor] = {}
    ops = []
    if rank == root:
        assert root_inputs is not None and len(root_inputs) == n_channels
        for c in range(n_channels):
            o = owner_of(c, world)
            if o == root:
                out[c] = root_inputs[c]
            else:
                ops.append(dist.P2POp(dist.isend, root_inputs[c], o, group=group, tag=c))
    else:
        for c in mine:
            out[c] = torch.empty(2 * samples, dtype=torch.float32, device=device)
            ops.append(dist.P2POp(dist.irecv, out[c], root, group=group, tag=c))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return out


def gather_outputs(outputs: Dict[int, torch.Tensor], n_channels: int, out_samples: int, device: torch.device,
                   root: int = 0, group=None) -> Optional[List[torch.Tensor]]:
    """Inverse of scatter_channels: root returns the list of per-channel outputs, other ranks None."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    ops = []
    result: Optional[List[torch.Tensor]] = None
    if rank == root:
        result = []
        for c in range(n_channels):
            o = owner_of(c, world)
            if o == root:
                result.append(outputs[c])
            else:
                t = torch.empty(2 * out_samples, dtype=torch.float32, device=device)
                result.append(t)
                ops.append(dist.P2POp(dist.irecv, t, o, group=group, tag=1000 + c))
    else:
        for c in sorted(outputs):
            ops.append(dist.P2POp(dist.isend, outputs[c], root, group=group, tag=1000 + c))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return result


def filter_channels(inputs: Dict[int, torch.Tensor], filter_fn: Callable[[int, torch.Tensor], torch.Tensor]
                    ) -> Dict[int, torch.Tensor]:
    """Run the per-channel filter (on a GPU rank: IfFir.process_device through the C-ABI) on every owned channel,
    back to back."""
    return {c: filter_fn(c, x) for c, x in sorted(inputs.items())}
