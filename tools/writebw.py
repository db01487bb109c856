"""compute-free yardsticks for the WRITE-heavy kernels of the path (the fused final stage writes 3 bytes for every byte it reads):
torch's fill (pure write), copy (1 read : 1 write) and a broadcast copy (1 read : 3 writes) over 1-2 GB, HIP events.
usage: python tools/writebw.py"""
import torch
dev = torch.device("cuda:0")
n = 512 << 20                      # elements of 2 bytes: 1 GiB
a = torch.empty(n, dtype=torch.float16, device=dev)
b = torch.empty(n, dtype=torch.float16, device=dev)
src = torch.empty(n // 3, dtype=torch.float16, device=dev)
dst3 = torch.empty((3, n // 3), dtype=torch.float16, device=dev)


def t(fn, reps=10):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3


gb = n * 2 / 1e9
print(f"fill   (write only)        {gb / t(lambda: a.fill_(1.0)) / 1e3:.2f} TB/s")
print(f"zero   (write only)        {gb / t(lambda: a.zero_()) / 1e3:.2f} TB/s")
print(f"copy   (1 read : 1 write)  {2 * gb / t(lambda: b.copy_(a)) / 1e3:.2f} TB/s of traffic")
g3 = (n // 3) * 2 / 1e9
print(f"bcast  (1 read : 3 writes) {4 * g3 / t(lambda: dst3.copy_(src[None].expand(3, -1))) / 1e3:.2f} TB/s of traffic")
print(f"read   (sum, read only)    {gb / t(lambda: a.view(torch.int16).max()) / 1e3:.2f} TB/s")
