"""times ph_dynconv_poolx / ph_dynconv / ph_pool at cfg2's part geometry (24 frames) -- same-box A/B of build variants (PH_ALT_LIB)"""
import sys, torch
sys.path.insert(0, ".")
from polyphonicformer_amd import _lib, engine as E
gpu = torch.device("cuda:0")
import os
B, N, H, W = int(os.environ.get('PH_PART_FRAMES', 32)), 153, 128, 256
HW, Npad = H * W, E.n_padded(N)
g = torch.Generator().manual_seed(1)
prec, feat = _lib.PH_PREC_BF16_KF16, _lib.PH_PREC_BF16
xp = (torch.randn(1, B, 256, HW, generator=g).to(gpu)).to(torch.bfloat16).view(torch.int16)
dp = xp.clone()
kern = (torch.randn(1, 2, B, Npad, 256, generator=g) * 0.1).to(gpu).to(torch.float16).view(torch.int16)
kbias = (torch.randn(2, B, Npad, generator=g) * 0.1).to(gpu)
bits = torch.zeros((B, Npad, HW // 32), dtype=torch.int32, device=gpu)
ns = max(1, 256 // B)              # the plan's split: one workgroup per CU
part = torch.zeros((B, ns, Npad, 512), device=gpu)
cnt = torch.zeros((B, ns, Npad), dtype=torch.int32, device=gpu)
part16 = torch.zeros((B, 16, Npad, 512), device=gpu)
def t(fn, n=30):
    for _ in range(60): fn()        # clocks ramp over the first milliseconds of a burst
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("poolx      %.1f us" % t(lambda: E.dynconv_poolx(xp, kern, kbias, N, HW, prec, bits, part)))
print("dynconv    %.1f us" % t(lambda: E.dynconv(xp, kern, kbias, 0, N, HW, prec, bits_out=bits)))
print("pool_depth %.1f us" % t(lambda: E.pool_depth_only(dp, bits, N, HW, feat, part, cnt)))
print("pool       %.1f us" % t(lambda: E.pool(xp, dp, bits, N, HW, feat, 16, out=part16)))
for n2 in (8, 10, 16, 21, 32):
    p2 = torch.zeros((B, n2, Npad, 512), device=gpu)
    print("poolx nsplit %d  %.1f us" % (n2, t(lambda: E.dynconv_poolx(xp, kern, kbias, N, HW, prec, bits, p2))))
    c2 = torch.zeros((B, n2, Npad), dtype=torch.int32, device=gpu)
    print("pool_depth nsplit %d  %.1f us" % (n2, t(lambda: E.pool_depth_only(dp, bits, N, HW, feat, p2, c2))))
for pr, nm in ((_lib.PH_PREC_F16, "f16"),):
    print("poolx %s  %.1f us" % (nm, t(lambda: E.dynconv_poolx(xp, kern, kbias, N, HW, pr, bits, part))))
    print("dynconv %s %.1f us" % (nm, t(lambda: E.dynconv(xp, kern, kbias, 0, N, HW, pr, bits_out=bits))))
