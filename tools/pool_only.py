"""runs only the pooling + conv (+ upsample, + fused conv-upsample) kernels at the cfg2 shape in the launch geometry of the headline step (24 frames per launch),
for the rocprofv3 --pmc passes.  usage: python tools/pool_only.py [mode = mixed16 | mixed | bf16 | fp16]"""
import sys, torch
sys.path.insert(0, ".")
from polyphonicformer_amd import _lib, engine as E
dev = torch.device("cuda:0")
mode = E.MODES[sys.argv[1] if len(sys.argv) > 1 else "mixed16"]
import os
N, B, H, W = 153, int(os.environ.get('PH_PART_FRAMES', 32)), 128, 256
HW = H * W
xp = torch.randint(-2**15, 2**15, (1, B, 256, E.hw_padded(HW)), dtype=torch.int16, device=dev) & 0x3BFF      # finite in bf16 and fp16
dp = xp.clone()
bits = torch.randint(-2**31, 2**31 - 1, (B, E.n_padded(N), E.hw_padded(HW) // 32), dtype=torch.int32, device=dev)
bits0 = bits.clone()
ns = E.default_nsplit(B, HW)
part = torch.empty((B, ns, E.n_padded(N), 512), dtype=torch.float32, device=dev)
cnt = torch.empty((B, ns, E.n_padded(N)), dtype=torch.int32, device=dev)
kern = torch.zeros((mode.KP, 2, B, 160, 256), dtype=torch.int16, device=dev)
kb = torch.zeros((2, B, 160), dtype=torch.float32, device=dev)
odt = torch.bfloat16 if mode.name == "bf16" else torch.float16
out = torch.empty((B, N, H, W), dtype=odt, device=dev)
up = torch.empty((B, N, 2 * H, 2 * W), dtype=odt, device=dev)
fused = mode.KP == 1 and _lib.load().ph_dynconv_up2_supported(N, H, W, mode.conv, E.OUT_CODE[odt])
px = bool(_lib.load().ph_dynconv_poolx_supported(N, mode.conv)) and mode.KP == 1
ns_px = 256 // B
part_px = torch.empty((B, ns_px, E.n_padded(N), 512), dtype=torch.float32, device=dev)
cnt_px = torch.empty((B, ns_px, E.n_padded(N)), dtype=torch.int32, device=dev)
for _ in range(5):
    if px:          # round 6: non-final conv + pooling of the x map in one kernel, then the pooling of depth_feats alone
        E.dynconv_poolx(xp, kern, kb, N, HW, mode.conv, bits, part_px)
        E.pool_depth_only(dp, bits, N, HW, mode.feat, part_px, cnt_px)
        bits.copy_(bits0)
    E.pool(xp, dp, bits, N, HW, mode.feat, ns, out=part, counts=cnt)
    E.dynconv(xp, kern, kb, 0, N, HW, mode.conv, bits_out=bits)
    E.dynconv(xp, kern, kb, 0, N, HW, mode.conv, logits_out=out, out_dtype=E.OUT_CODE[odt])
    E.upsample2x(out, out=up)
    if fused:       # round 4: the final stage's conv + x2 upsample in one kernel (mask form with low-res logits, depth form without)
        E.dynconv_up2(xp, kern, kb, 0, N, H, W, mode.conv, up, logits_out=out, out_dtype=E.OUT_CODE[odt])
        E.dynconv_up2(dp, kern, kb, 1, N, H, W, mode.conv, up, logits_out=None, out_dtype=E.OUT_CODE[odt])
torch.cuda.synchronize()
