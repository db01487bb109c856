import torch, time
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n
a = torch.empty(1<<28, dtype=torch.float32, device='cuda')  # 1 GiB
b = torch.empty_like(a)
ms = t(lambda: a.fill_(1.0)); print("fill  1GiB: %.3f ms -> %.2f TB/s" % (ms, a.numel()*4/ms/1e9))
ms = t(lambda: b.copy_(a)); print("copy  1GiB: %.3f ms -> %.2f TB/s (r+w)" % (ms, 2*a.numel()*4/ms/1e9))
ms = t(lambda: a.sum()); print("read  1GiB: %.3f ms -> %.2f TB/s" % (ms, a.numel()*4/ms/1e9))
h = a.view(torch.bfloat16)[:1<<27]
