import sys, torch
sys.path.insert(0, ".")
from polyphonicformer_amd import _lib
from bench import time_op
lib = _lib.load()
a = torch.randint(0, 2**31 - 1, (1 << 28,), dtype=torch.int32, device="cuda")   # 1 GiB
out = torch.zeros(4, dtype=torch.int32, device="cuda")
for blocks in (1024, 2048, 4096, 8192, 16384):
    t = time_op(lambda: lib.ph_selftest_readbw(_lib.ptr(a), a.numel() * 4, blocks, _lib.ptr(out), _lib.stream_ptr()), 10)
    print(f"read 1 GiB, {blocks} blocks: {t:.3f} ms = {a.numel()*4/t/1e9:.2f} TB/s")
b = a[: (1 << 26)]
t = time_op(lambda: lib.ph_selftest_readbw(_lib.ptr(b), b.numel() * 4, 4096, _lib.ptr(out), _lib.stream_ptr()), 10)
print(f"read 256 MiB (fits the Infinity Cache?): {t:.3f} ms = {b.numel()*4/t/1e9:.2f} TB/s")
