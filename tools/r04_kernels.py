"""GPU, round 4: isolated timings (HIP events, launches back to back on one stream) of the a6 kernels at the headline geometry
(cfg2, 32 frames per launch (PH_PART_FRAMES; 24 until round 6), `mixed16` / `fp16` arithmetic) -- the two-kernel final stage against the fused conv + x2 upsample
(ph_dynconv_up2) -- and of a1 (16 frames).  usage: python tools/r04_kernels.py [mode]"""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polyphonicformer_amd import _lib, engine as E
if os.environ.get("PH_ALT_LIB"):
    _lib.LIB_PATH = os.environ["PH_ALT_LIB"]            # same-box A/B of a library variant
dev = torch.device("cuda:0")
mode = E.MODES[sys.argv[1] if len(sys.argv) > 1 else "mixed16"]
N, B, H, W = 153, int(os.environ.get("R04_B", os.environ.get("PH_PART_FRAMES", "32"))), 128, 256
HW = H * W
g = torch.Generator(device="cpu").manual_seed(1)
pdt = mode.feat_dtype
xp = torch.randn(B, 256, HW, generator=g).to(pdt).view(torch.int16)[None].contiguous().to(dev)
dp = xp.clone()
bits = torch.randint(-2**31, 2**31 - 1, (B, E.n_padded(N), HW // 32), dtype=torch.int32, device=dev)
ns = E.default_nsplit(B, HW)
part = torch.empty((B, ns, E.n_padded(N), 512), dtype=torch.float32, device=dev)
cnt = torch.empty((B, ns, E.n_padded(N)), dtype=torch.int32, device=dev)
kdt = torch.bfloat16 if mode.name == "bf16" else torch.float16
kern = (torch.randn(1, 2, B, 160, 256, generator=g) * 0.1).to(kdt).view(torch.int16).contiguous().to(dev)
kb = torch.zeros((2, B, 160), dtype=torch.float32, device=dev)
odt = torch.bfloat16 if mode.name == "bf16" else torch.float16
oc = E.OUT_CODE[odt]
low = torch.empty((B, N, H, W), dtype=odt, device=dev)
up = torch.empty((B, N, 2 * H, 2 * W), dtype=odt, device=dev)
up_b = torch.empty_like(up)


def t(fn, n=20):
    for _ in range(40):          # the clocks ramp over the first milliseconds of a burst
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3      # us


feat = B * 256 * HW * 2
res = {"mode": mode.name, "frames": B}
res["pool_us"] = t(lambda: E.pool(xp, dp, bits, N, HW, mode.feat, ns, out=part, counts=cnt))
bits2 = bits.clone()
res["dynconv_bits_us"] = t(lambda: E.dynconv(xp, kern, kb, 0, N, HW, mode.conv, bits_out=bits2))
res["dynconv_logits_us"] = t(lambda: E.dynconv(xp, kern, kb, 0, N, HW, mode.conv, logits_out=low, out_dtype=oc))
res["upsample2x_us"] = t(lambda: E.upsample2x(low, out=up))


def two():
    E.dynconv(xp, kern, kb, 0, N, HW, mode.conv, logits_out=low, out_dtype=oc)
    E.upsample2x(low, out=up)


res["two_kernel_final_branch_us"] = t(two)
if _lib.load().ph_dynconv_up2_supported(N, H, W, mode.conv, oc):
    res["up2_mask_us"] = t(lambda: E.dynconv_up2(xp, kern, kb, 0, N, H, W, mode.conv, up_b, logits_out=low, out_dtype=oc))
    res["up2_depth_us"] = t(lambda: E.dynconv_up2(dp, kern, kb, 1, N, H, W, mode.conv, up_b, logits_out=None, out_dtype=oc))
    bm = feat + B * N * HW * 2 * 5
    bd = feat + B * N * HW * 2 * 4
    res["up2_mask_TBps"] = bm / res["up2_mask_us"] / 1e6
    res["up2_depth_TBps"] = bd / res["up2_depth_us"] / 1e6
if mode.KP == 1 and _lib.load().ph_dynconv_poolx_supported(N, mode.conv):       # round 6
    ns_px = 256 // B
    part_px = torch.empty((B, ns_px, E.n_padded(N), 512), dtype=torch.float32, device=part.device)
    cnt_px = torch.empty((B, ns_px, E.n_padded(N)), dtype=torch.int32, device=part.device)
    res["dynconv_poolx_us"] = t(lambda: E.dynconv_poolx(xp, kern, kb, N, HW, mode.conv, bits2, part_px))
    res["pool_depth_us"] = t(lambda: E.pool_depth_only(dp, bits2, N, HW, mode.feat, part_px, cnt_px))
    res["dynconv_poolx_TBps"] = (feat + B * N * HW // 8) / res["dynconv_poolx_us"] / 1e6
    res["pool_depth_TBps"] = (feat + B * N * HW // 8) / res["pool_depth_us"] / 1e6
res["pool_TBps"] = (2 * feat + B * N * HW // 8) / res["pool_us"] / 1e6
res["dynconv_bits_TBps"] = (feat + B * N * HW // 8) / res["dynconv_bits_us"] / 1e6
res["dynconv_logits_TBps"] = (feat + B * N * HW * 2) / res["dynconv_logits_us"] / 1e6
res["upsample2x_TBps"] = (B * N * HW * 2 * 5) / res["upsample2x_us"] / 1e6
print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in res.items()}), flush=True)
