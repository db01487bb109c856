"""runs only the pooling + conv kernels at the cfg2 shape (for rocprofv3 --pmc passes)"""
import sys, torch
sys.path.insert(0, ".")
from polyphonicformer_amd import _lib, engine as E
dev = torch.device("cuda:0")
N, B, H, W = 153, 24, 128, 256
HW = H * W
xp = torch.randint(-2**15, 2**15, (1, B, 256, E.hw_padded(HW)), dtype=torch.int16, device=dev) & 0x3FFF
dp = xp.clone()
bits = torch.randint(-2**31, 2**31 - 1, (B, E.n_padded(N), E.hw_padded(HW) // 32), dtype=torch.int32, device=dev)
ns = 5
part = torch.empty((B, ns, E.n_padded(N), 512), dtype=torch.float32, device=dev)
kern = torch.zeros((1, 2, B, 160, 256), dtype=torch.int16, device=dev)
kb = torch.zeros((2, B, 160), dtype=torch.float32, device=dev)
out = torch.empty((B, N, H, W), dtype=torch.bfloat16, device=dev)
for _ in range(5):
    E.pool(xp, dp, bits, N, HW, 1, ns, out=part)
    E.dynconv(xp, kern, kb, 0, N, HW, 1, bits_out=bits)
    E.dynconv(xp, kern, kb, 0, N, HW, 1, logits_out=out, out_dtype=1)
torch.cuda.synchronize()
