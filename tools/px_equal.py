import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from polyphonicformer_amd import _lib, engine as E
gpu = torch.device("cuda:0")
def planes16(t, dt): return t.to(dt).view(torch.int16)
for prec, feat, kdt in ((_lib.PH_PREC_F16, _lib.PH_PREC_F16, torch.float16), (_lib.PH_PREC_BF16, _lib.PH_PREC_BF16, torch.bfloat16), (_lib.PH_PREC_BF16_KF16, _lib.PH_PREC_BF16, torch.float16)):
    for (N, H, W, B, ns) in ((153, 128, 256, 2, 32), (153, 128, 256, 3, 8), (111, 32, 64, 2, 4), (153, 48, 156, 2, 7)):
        g = torch.Generator().manual_seed(7 + N + H)
        HW, Npad = H * W, E.n_padded(N)
        x, d = torch.randn(B, 256, H, W, generator=g), torch.randn(B, 256, H, W, generator=g)
        kern = planes16(torch.randn(2, B, Npad, 256, generator=g) * 0.1, kdt)[None].contiguous().to(gpu)
        kbias = (torch.randn(2, B, Npad, generator=g) * 0.1).to(gpu)
        xp, dp = E.ingest(x.to(gpu), feat), E.ingest(d.to(gpu), feat)
        HWp = E.hw_padded(HW)
        bits_ref = torch.zeros((B, Npad, HWp // 32), dtype=torch.int32, device=gpu)
        E.dynconv(xp, kern, kbias, 0, N, HW, prec, bits_out=bits_ref)
        part_ref = torch.zeros((B, ns, Npad, 512), device=gpu); cnt_ref = torch.zeros((B, ns, Npad), dtype=torch.int32, device=gpu)
        E.pool(xp, dp, bits_ref, N, HW, feat, ns, out=part_ref, counts=cnt_ref)
        bits = torch.zeros_like(bits_ref); part = torch.zeros_like(part_ref)
        E.dynconv_poolx(xp, kern, kbias, N, HW, prec, bits, part)
        same = torch.equal(part[..., :256][:, :, :N], part_ref[..., :256][:, :, :N])
        diff = (part[..., :256] - part_ref[..., :256]).abs().max().item()
        print(prec, (N, H, W, B, ns), "x columns bit-identical per split:", same, "max abs diff", diff)
