import torch
for n in (226_000_000, 113_000_000):
    a=torch.randn(n,device='cuda').bfloat16(); b=torch.randn(n,device='cuda').bfloat16(); c=torch.empty_like(a)
    for _ in range(3): torch.add(a,b,out=c)
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): torch.add(a,b,out=c)
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/10
    print(n, 'add: %.3f ms  %.2f TB/s' % (ms, 3*n*2/ms/1e9))
    e0.record()
    for _ in range(10): c.copy_(a)
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/10
    print(n, 'copy: %.3f ms  %.2f TB/s' % (ms, 2*n*2/ms/1e9))
