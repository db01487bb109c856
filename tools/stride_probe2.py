"""is the 64 KB (power-of-two) channel-row stride of cfg2's planes a DRAM channel / bank problem?  pool and conv-bits at H*W = 32768 against
the neighbouring non-power-of-two strides (same bytes to 0.4 %), 24 frames.  usage: python tools/stride_probe2.py"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polyphonicformer_amd import _lib, engine as E
from bench import time_op
dev = torch.device("cuda:0")
N, B = 153, 24
for (H, W) in [(128, 256), (128, 255), (128, 257), (128, 254), (128, 256), (127, 256), (129, 256)]:
    HW = H * W
    xp = (torch.randint(-2**15, 2**15, (1, B, 256, E.hw_padded(HW)), dtype=torch.int16, device=dev) & 0x3BFF)
    dp = xp.clone()
    bits = torch.randint(-2**31, 2**31 - 1, (B, E.n_padded(N), E.hw_padded(HW) // 32), dtype=torch.int32, device=dev)
    ns = E.default_nsplit(B, HW)
    part = torch.empty((B, ns, E.n_padded(N), 512), dtype=torch.float32, device=dev)
    cnt = torch.empty((B, ns, E.n_padded(N)), dtype=torch.int32, device=dev)
    t = time_op(lambda: E.pool(xp, dp, bits, N, HW, _lib.PH_PREC_BF16, ns, out=part, counts=cnt), 20)
    byts = 2 * B * 256 * HW * 2
    kern = torch.zeros((1, 2, B, 160, 256), dtype=torch.int16, device=dev)
    kb = torch.zeros((2, B, 160), dtype=torch.float32, device=dev)
    t2 = time_op(lambda: E.dynconv(xp, kern, kb, 0, N, HW, _lib.PH_PREC_BF16_KF16, bits_out=bits), 20)
    print(f"H x W = {H} x {W}: row stride {E.hw_padded(HW) * 2} B  pool {t * 1e3:.1f} us = {byts / t / 1e9:.2f} TB/s | conv bits {t2 * 1e3:.1f} us = {byts / 2 / t2 / 1e9:.2f} TB/s", flush=True)
