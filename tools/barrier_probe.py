#!/usr/bin/env python3
"""barrier_probe.py — how long RCCL's barrier / all_reduce take through torch.distributed with one rank on this box (development
tool): the first collective of a process sets the transport up (6.5 ms on MI355X), later ones take ~30 us.  bench.py spends the
first ones right after init_process_group for that reason."""
import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29512")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
for i in range(6):
    torch.cuda.synchronize(); t0=time.perf_counter(); dist.barrier(); torch.cuda.synchronize(); print("barrier %d: %.3f ms"%(i,(time.perf_counter()-t0)*1e3))
t=torch.zeros(3,dtype=torch.float64,device="cuda")
for i in range(3):
    torch.cuda.synchronize(); t0=time.perf_counter(); dist.all_reduce(t,op=dist.ReduceOp.MAX); torch.cuda.synchronize(); print("all_reduce %d: %.3f ms"%(i,(time.perf_counter()-t0)*1e3))
dist.destroy_process_group()
