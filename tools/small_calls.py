import sys, time
sys.path.insert(0,'/root/repo')
import torch, __graft_entry__ as g
fir=g.load_pkg().if_fir
torch.cuda.set_device(0)
taps=fir.bpf_design(255)
for log2n in (16,18,20,22,24):
    n=1<<log2n
    x=torch.empty(2*n,dtype=torch.float32,device='cuda'); 
    with fir.IfFir(taps,4,0, dev=True) as f:
        y=torch.empty(2*f.out_count(n),dtype=torch.float32,device='cuda')
        f.synth_device(x.data_ptr(),0,n,0); f.synchronize()
        for _ in range(20): f.process_device(x.data_ptr(),y.data_ptr(),n)
        f.synchronize()
        reps=2000 if log2n<22 else 300
        t0=time.perf_counter()
        for _ in range(reps): f.process_device(x.data_ptr(),y.data_ptr(),n)
        t_issue=time.perf_counter()-t0
        f.synchronize()
        t=time.perf_counter()-t0
        print("n=2^%d: %.2f us per call (host issue %.2f us) -> %.1f GS/s"%(log2n,t/reps*1e6,t_issue/reps*1e6,n*reps/t/1e9))
