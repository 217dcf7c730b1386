import torch, time
x = torch.empty(100*1024*1024, dtype=torch.float32, device="cuda").normal_()
y = torch.empty_like(x)
for name, fn, bytes_ in (("sum 400MB", lambda: x.sum(), x.numel()*4), ("copy 400MB", lambda: y.copy_(x), 2*x.numel()*4), ("mul 400MB", lambda: torch.mul(x, 2.0, out=y), 2*x.numel()*4)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/20
    print("%s: %.3f ms, %.2f TB/s" % (name, ms, bytes_/ms/1e9))
