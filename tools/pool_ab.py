"""k_pool alone at the headline's launch geometry (cfg2, 24 frames, nsplit 5) and at 16 / 32 frames: run under PH_POOL_XCD=0 / 1
(3-D grid / XCD-aware 1-D grid, read once per process) for the same-box A/B.  usage: python tools/pool_ab.py [mode]"""
import os, sys, torch
sys.path.insert(0, ".")
from polyphonicformer_amd import engine as E
from bench import time_op
dev = torch.device("cuda:0")
mode = E.MODES[sys.argv[1] if len(sys.argv) > 1 else "mixed16"]
N, H, W = 153, 128, 256
HW = H * W
for B in (24, 16, 32):
    xp = torch.randint(-2**15, 2**15, (1, B, 256, E.hw_padded(HW)), dtype=torch.int16, device=dev) & 0x3BFF
    dp = xp.clone()
    bits = torch.randint(-2**31, 2**31 - 1, (B, E.n_padded(N), E.hw_padded(HW) // 32), dtype=torch.int32, device=dev)
    ns = E.default_nsplit(B, HW)
    part = torch.empty((B, ns, E.n_padded(N), 512), dtype=torch.float32, device=dev)
    cnt = torch.empty((B, ns, E.n_padded(N)), dtype=torch.int32, device=dev)
    ts = [time_op(lambda: E.pool(xp, dp, bits, N, HW, mode.feat, ns, out=part, counts=cnt), 30, warm=3) for _ in range(3)]
    byt = B * (2 * 256 * HW * 2 + N * HW // 8)
    print(f"PH_POOL_XCD={os.environ.get('PH_POOL_XCD', '1')} B={B} nsplit={ns}: " + " ".join(f"{t*1e3:.1f}" for t in ts) + f" us  best {byt/min(ts)/1e9:.2f} TB/s")
