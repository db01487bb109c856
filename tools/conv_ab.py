"""times k_conv_nhwc alone (3x3 stride 1 at 128x256, 3x3 stride 2 at 256x512 -> 128x256, 1x1 at 128x256; 16 frames, fp16 grade) for
same-box A/B of library variants.  usage: python tools/conv_ab.py [path of an alternative libpolyhead.so | -] [B] [c16]"""
import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polyphonicformer_amd import _lib
if len(sys.argv) > 1 and sys.argv[1] != "-":
    _lib.LIB_PATH = sys.argv[1]
from polyphonicformer_amd import engine as E
from polyphonicformer_amd.pack import pack_b32
from bench import time_op
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device("cuda:0")
prec = _lib.PH_PREC_F16
lib = _lib.load()
g = torch.Generator().manual_seed(3)
out = []
for (k, s, H, W) in ((3, 1, 128, 256), (3, 2, 256, 512), (3, 1, 64, 128)):
    x = (torch.randn(1, B, H * W, 256, generator=g).to(torch.float16).view(torch.int16)).to(dev)
    w = torch.randn(256, k * k * 256, generator=g, dtype=torch.float64) * 0.02
    wp = pack_b32(E._planes_of(w, 1, True)[0])[None].contiguous().to(dev)
    Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
    y = torch.empty((B, Ho * Wo, 256), dtype=torch.float32, device=dev)
    partial = torch.zeros((lib.ph_conv_nhwc_partial_floats(B, Ho, Wo),), dtype=torch.float32, device=dev)
    pk = dict(wp=wp, k=k, s=s)
    lay = _lib.PH_PLANES_C16 if (s == 2 and len(sys.argv) > 3 and sys.argv[3] == "c16") else 0   # timing only: the same bytes read as chunk-major
    ts = [time_op(lambda: E.conv_nhwc(x, pk, y, partial, B, H, W, prec | lay), 10, warm=2) for _ in range(3)]
    fl = 2.0 * B * Ho * Wo * 256 * k * k * 256
    out.append(f"{k}x{k}s{s}@{H}x{W}: {min(ts)*1e3:.0f} us {fl / (min(ts) * 1e-3) / 1e12:.0f} TF/s")
print(_lib.LIB_PATH.split('/')[-1], "|", " | ".join(out), "| checksum", float(y.double().sum()))
