"""dynamic-conv time against the number of query rows (waves per workgroup), modes bf16 and mixed: is the mixed mode's 2-MFMA form
bound by the SIMD that hosts two of its five waves?  usage: python tools/conv_rows.py"""
import sys, json, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from polyphonicformer_amd import _lib, engine as E
dev = torch.device("cuda:0")
B, H, W = 24, 128, 256
HW = H * W
for name in ("bf16", "mixed"):
    mode = E.MODES[name]
    xp = torch.randint(-2**15, 2**15, (1, B, 256, E.hw_padded(HW)), dtype=torch.int16, device=dev) & 0x3BFF
    odt = torch.bfloat16 if name == "bf16" else torch.float16
    for N in (64, 96, 128, 153, 192, 253):
        Np = E.n_padded(N)
        bits = torch.zeros((B, Np, E.hw_padded(HW) // 32), dtype=torch.int32, device=dev)
        kern = torch.zeros((mode.KP, 2, B, Np, 256), dtype=torch.int16, device=dev)
        kb = torch.zeros((2, B, Np), dtype=torch.float32, device=dev)
        out = torch.empty((B, N, H, W), dtype=odt, device=dev)
        tb = bench.time_op(lambda: E.dynconv(xp, kern, kb, 0, N, HW, mode.conv, bits_out=bits), 10) * 1e3
        tl = bench.time_op(lambda: E.dynconv(xp, kern, kb, 0, N, HW, mode.conv, logits_out=out, out_dtype=E.OUT_CODE[odt]), 10) * 1e3
        print(json.dumps({"mode": name, "N": N, "row_blocks": Np // 32, "bits_us": round(tb, 1), "logits_us": round(tl, 1)}), flush=True)
        del out
