"""Does the 256 MiB Infinity Cache serve the feature planes when a launch covers few frames?  Times k_pool (the path's dominant,
HBM-bound kernel) at 1..24 frames per launch, launched back to back on the SAME buffers (a working set below the cache size
stays resident if the cache keeps it), once with the library's nt|sc1 stream policy and once with a library built with
-DPH_CPOL_STREAM=0.  usage: python tools/mall_probe.py [path of an alternative libpolyhead.so]"""
import sys, torch
sys.path.insert(0, ".")
from polyphonicformer_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = sys.argv[1]
from polyphonicformer_amd import engine as E
from bench import time_op
dev = torch.device("cuda:0")
mode = E.MODES["mixed16"]
N, H, W = 153, 128, 256
HW = H * W
print("library:", _lib.LIB_PATH)
for B in (1, 2, 3, 4, 6, 8, 12, 24):
    xp = torch.randint(-2**15, 2**15, (1, B, 256, E.hw_padded(HW)), dtype=torch.int16, device=dev) & 0x3BFF
    dp = xp.clone()
    bits = torch.randint(-2**31, 2**31 - 1, (B, E.n_padded(N), E.hw_padded(HW) // 32), dtype=torch.int32, device=dev)
    for ns in sorted({E.default_nsplit(B, HW), min(64, max(1, 1024 // (4 * B)))}):
        part = torch.empty((B, ns, E.n_padded(N), 512), dtype=torch.float32, device=dev)
        cnt = torch.empty((B, ns, E.n_padded(N)), dtype=torch.int32, device=dev)
        t = time_op(lambda: E.pool(xp, dp, bits, N, HW, mode.feat, ns, out=part, counts=cnt), 20, warm=3)
        byt = B * (2 * 256 * HW * 2 + N * HW // 8)
        print(f"B={B:2d} nsplit={ns:2d} wgs={4*B*ns:4d}  {t*1e3:7.1f} us  {t*1e3/B:6.2f} us/frame  {byt/t/1e9:6.2f} TB/s  working set {byt/1e6:.0f} MB")
