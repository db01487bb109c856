"""is the pool / conv kernel limited by the 64 KB channel-row stride (TLB / DRAM page locality)?
same total bytes, different H*W (row stride) -- tuning probe"""
import sys, torch
sys.path.insert(0, ".")
from polyphonicformer_amd import _lib, engine as E
from bench import time_op
dev = torch.device("cuda:0")
N = 153
for (B, H, W) in [(24, 128, 256), (96, 64, 128), (384, 32, 64), (1536, 16, 32)]:
    HW = H * W
    xp = torch.randint(-2**15, 2**15, (1, B, 256, E.hw_padded(HW)), dtype=torch.int16, device=dev) & 0x3FFF
    dp = xp.clone()
    bits = torch.randint(-2**31, 2**31 - 1, (B, E.n_padded(N), E.hw_padded(HW) // 32), dtype=torch.int32, device=dev)
    ns = max(1, min(512 // (4 * B), E.hw_padded(HW) // 128))
    part = torch.empty((B, ns, E.n_padded(N), 512), dtype=torch.float32, device=dev)
    t = time_op(lambda: E.pool(xp, dp, bits, N, HW, 1, ns, out=part), 10)
    byts = 2 * B * 256 * HW * 2
    kern = torch.zeros((1, 2, B, 160, 256), dtype=torch.int16, device=dev)
    kb = torch.zeros((2, B, 160), dtype=torch.float32, device=dev)
    t2 = time_op(lambda: E.dynconv(xp, kern, kb, 0, N, HW, 1, bits_out=bits), 10)
    print(f"B={B} HW={HW} stride={HW*2/1024:.0f}KB nsplit={ns}: pool {t:.3f} ms = {byts/t/1e9:.2f} TB/s | conv_bits {t2:.3f} ms = {byts/2/t2/1e9:.2f} TB/s")
