import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, __graft_entry__ as g
fir=g.load_pkg().if_fir
torch.cuda.set_device(0)
n=1<<28
x=torch.empty(2*n,dtype=torch.float32,device='cuda')
with fir.IfFir(fir.bpf_design(255),4,0) as f:
    f.synth_device(x.data_ptr(),0,n,0); f.synchronize()
    for _ in range(3): f.power_device(x.data_ptr(), n)
    t0=time.perf_counter()
    for _ in range(20): p=f.power_device(x.data_ptr(), n)
    dt=(time.perf_counter()-t0)/20
    print("power of 2^28 samples: %.4f ms per call (incl. sync + 8-byte readback) = %.0f GB/s; mean power %.6f"%(dt*1e3, 8*n/dt/1e9, p))
