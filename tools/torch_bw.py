import torch
torch.cuda.set_device(0)
y=torch.empty(1<<29,dtype=torch.float32,device='cuda')
x=torch.empty(1<<29,dtype=torch.float32,device='cuda')
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
for name,fn,bytes_ in (("fill (write only)", lambda: y.fill_(1.0), 4*(1<<29)), ("sum (read only)", lambda: x.sum(), 4*(1<<29)), ("copy", lambda: y.copy_(x), 8*(1<<29))):
    for _ in range(3): fn()
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/10
    print("%s: %.3f ms -> %.0f GB/s"%(name,ms,bytes_/ms/1e6))
