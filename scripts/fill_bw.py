import torch, time
x=torch.empty(1<<28, dtype=torch.float32, device="cuda")   # 1 GiB
y=torch.empty(1<<28, dtype=torch.float32, device="cuda")
for name,fn,bytes_ in (("fill",lambda: x.fill_(1.0), x.numel()*4), ("zero",lambda: x.zero_(), x.numel()*4), ("copy",lambda: y.copy_(x), 2*x.numel()*4), ("read(sum)",lambda: x.sum(), x.numel()*4)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/10
    print(name, "%.3f ms"%ms, "%.0f GB/s"%(bytes_/ms/1e6))
