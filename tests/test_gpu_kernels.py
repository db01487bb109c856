"""GPU: every libpolyhead kernel in isolation against a plain torch fp32/fp64 reference of the same
op, called through the C ABI (ctypes).  Tolerances are stated per test."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import helpers as Hh
from polyphonicformer_amd import _lib, engine as E

pytestmark = pytest.mark.gpu
PRECS = [_lib.PH_PREC_BF16, _lib.PH_PREC_SPLIT]


def bf16_bits(t):
    return t.to(torch.bfloat16).view(torch.int16)


def planes_to_float(pl):
    """int16 planes [P, ...] -> float64 sum of planes"""
    return sum(pl[p].view(torch.bfloat16).double() for p in range(pl.shape[0]))


def unpack_bits(bits, N, HW):
    B, Npad, Wd = bits.shape
    b = bits.cpu().numpy().view(np.uint32)
    out = np.unpackbits(b.view(np.uint8).reshape(B, Npad, Wd, 4), axis=-1, bitorder="little")
    return torch.from_numpy(out.reshape(B, Npad, Wd * 32)[:, :N, :HW].astype(np.float32))


def test_selftest_mfma_layouts(gpu):
    lib = _lib.load()
    g = torch.Generator().manual_seed(0)
    a = torch.randn(16, 32, generator=g).to(torch.bfloat16)
    b = torch.randn(32, 16, generator=g).to(torch.bfloat16)      # asymmetric B
    d = torch.zeros(16, 16, device=gpu)
    ad, bd = a.view(torch.int16).to(gpu), b.t().contiguous().view(torch.int16).to(gpu)   # keep alive
    _lib.check(lib.ph_selftest_mfma16(_lib.ptr(ad), _lib.ptr(bd), _lib.ptr(d), _lib.stream_ptr()), "mfma16")
    ref = a.double() @ b.double()
    assert (d.cpu().double() - ref).abs().max() < 1e-4
    a = torch.randn(32, 16, generator=g).to(torch.bfloat16)
    b = torch.randn(16, 32, generator=g).to(torch.bfloat16)
    d = torch.zeros(32, 32, device=gpu)
    ad, bd = a.view(torch.int16).to(gpu), b.t().contiguous().view(torch.int16).to(gpu)
    _lib.check(lib.ph_selftest_mfma32(_lib.ptr(ad), _lib.ptr(bd), _lib.ptr(d), _lib.stream_ptr()), "mfma32")
    ref = a.double() @ b.double()
    assert (d.cpu().double() - ref).abs().max() < 1e-4


def test_selftest_transposing_lds_read(gpu):
    lib = _lib.load()
    src = torch.arange(256, dtype=torch.int16)
    out = torch.zeros(64, 4, dtype=torch.int16, device=gpu)
    srcd = src.to(gpu)
    _lib.check(lib.ph_selftest_trread(_lib.ptr(srcd), _lib.ptr(out), _lib.stream_ptr()), "trread")
    out = out.cpu().numpy()
    exp = np.zeros((64, 4), dtype=np.int16)
    for l in range(64):
        g, i = l >> 4, l & 15
        for j in range(4):
            exp[l, j] = (4 * g + j) * 16 + i
    assert np.array_equal(out, exp), f"ds_read_tr16_b64 semantics differ:\n{out}"


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("H,W", [(8, 16), (6, 13)])
def test_ingest(gpu, prec, H, W):
    x = torch.randn(2, 256, H, W, generator=torch.Generator().manual_seed(1))
    pl = E.ingest(x.to(gpu), prec).cpu()
    HW = H * W
    rec = planes_to_float(pl)[..., :HW].reshape(2, 256, H, W)
    tol = 2.0 ** -8 if prec == _lib.PH_PREC_BF16 else 2.0 ** -16
    assert ((rec - x.double()).abs() <= tol * x.double().abs() + 1e-30).all()
    assert (planes_to_float(pl)[..., HW:] == 0).all()        # zero padding
    assert torch.equal(pl[0, :, :, :HW].reshape(2, 256, H, W), bf16_bits(x))   # hi plane == RNE bf16


@pytest.mark.parametrize("N,H,W", [(111, 8, 16), (7, 6, 13), (153, 16, 24)])
def test_binarize(gpu, N, H, W):
    m = torch.randn(2, N, H, W, generator=torch.Generator().manual_seed(2))
    # boundary of the reference's definition `sigmoid(m) > 0.5` in fp32: true from 97 * 2^-30 on (1.5 * 2^-24 is the
    # last value whose sigmoid still rounds to 0.5)
    m[0, 0, 0, :8] = torch.tensor([0.0, -0.0, 1e-30, -1e-30, 96 * 2.0 ** -30, 97 * 2.0 ** -30, 2.0 ** -24, 2.0 ** -23])
    bits = E.binarize(m.to(gpu))
    got = unpack_bits(bits, N, H * W)
    assert torch.equal(got, (m.sigmoid() > 0.5).float().reshape(2, N, -1))       # kernel_update_head.py:236-238
    assert got[0, 0, :8].tolist() == [0, 0, 0, 0, 0, 1, 0, 1]
    full = unpack_bits(bits, bits.shape[1], bits.shape[2] * 32)
    assert full[:, N:].sum() == 0 and full[:, :, H * W:].sum() == 0


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("N,H,W", [(111, 8, 16), (40, 6, 13), (153, 16, 32)])
def test_binarize_16bit_logits(gpu, dt, N, H, W):
    """round 5: mask logits that arrive 16-bit (a 16-bit KernelHead grade's hand-over, cfg2's bf16 inputs) are binarised from their
    own format -- exactly `sigmoid(m.float()) > 0.5` of the 16-bit values, ragged sizes (scalar path) included"""
    m = torch.randn(2, N, H, W, generator=torch.Generator().manual_seed(4)).to(dt)
    m[0, 0, 0, :6] = torch.tensor([0.0, -0.0, 6e-8, -6e-8, 2.0 ** -14, -2.0 ** -14]).to(dt)
    bits = E.binarize(m.to(gpu))
    got = unpack_bits(bits, N, H * W)
    assert torch.equal(got, (m.float().sigmoid() > 0.5).float().reshape(2, N, -1))
    assert torch.equal(got, unpack_bits(E.binarize(m.float().to(gpu)), N, H * W))      # = the fp32 kernel on the same values


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("N,H,W,nsplit", [(111, 8, 16, 1), (153, 16, 32, 3), (40, 6, 13, 1), (253, 8, 48, 2)])
def test_pool(gpu, prec, N, H, W, nsplit):
    g = torch.Generator().manual_seed(3)
    B, HW = 2, H * W
    x, d = torch.randn(B, 256, H, W, generator=g), torch.randn(B, 256, H, W, generator=g)
    m = torch.randn(B, N, H, W, generator=g)
    xp, dp = E.ingest(x.to(gpu), prec), E.ingest(d.to(gpu), prec)
    bits = E.binarize(m.to(gpu))
    part = E.pool(xp, dp, bits, N, HW, prec, nsplit=nsplit).cpu().double()
    got = part.sum(1)[:, :N]
    M = (m.sigmoid() > 0.5).double().reshape(B, N, HW)
    xq, dq = planes_to_float(xp.cpu())[..., :HW], planes_to_float(dp.cpu())[..., :HW]
    ref = torch.cat([torch.einsum("bnk,bck->bnc", M, xq), torch.einsum("bnk,bck->bnc", M, dq)], -1)
    # exact products, fp32 accumulation over <= HW terms
    assert (got - ref).abs().max() <= 1e-5 * ref.abs().max()
    assert part.sum(1)[:, N:].abs().max() == 0
    # vs the unquantised fp32 features: bf16 rounding (2^-9 rel per element) / split (2^-17)
    ref32 = torch.cat([torch.einsum("bnk,bck->bnc", M, x.double().reshape(B, 256, HW)),
                       torch.einsum("bnk,bck->bnc", M, d.double().reshape(B, 256, HW))], -1)
    tol = 4e-3 if prec == _lib.PH_PREC_BF16 else 2e-5
    assert Hh.rel_err(got, ref32) < tol
    # x-only variant (KernelHead pooling)
    part1 = E.pool(xp, None, bits, N, HW, prec, nsplit=nsplit).cpu().double().sum(1)[:, :N, :256]
    assert (part1 - ref[..., :256]).abs().max() <= 1e-5 * ref.abs().max()


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("N,H,W", [(111, 8, 16), (153, 16, 32), (40, 6, 13), (253, 8, 48)])
def test_dynconv(gpu, prec, N, H, W):
    g = torch.Generator().manual_seed(4)
    B, HW = 2, H * W
    P = 2 if prec == _lib.PH_PREC_SPLIT else 1
    Npad = E.n_padded(N)
    x = torch.randn(B, 256, H, W, generator=g)
    kern_f = torch.randn(2, B, Npad, 256, generator=g) * 0.1
    kbias = torch.randn(2, B, Npad, generator=g) * 0.1
    hi = kern_f.to(torch.bfloat16)
    planes = [hi.view(torch.int16)]
    if P == 2:
        planes.append((kern_f - hi.float()).to(torch.bfloat16).view(torch.int16))
    kern = torch.stack(planes, 0).contiguous().to(gpu)
    xp = E.ingest(x.to(gpu), prec)
    kq = planes_to_float(kern.cpu())
    kbias_d = kbias.to(gpu)
    xq = planes_to_float(xp.cpu())[..., :HW]
    for br in (0, 1):
        ref = torch.einsum("bnc,bck->bnk", kq[br, :, :N], xq) + kbias[br, :, :N, None].double()
        out = torch.empty(B, N, H, W, device=gpu)
        E.dynconv(xp, kern, kbias_d, br, N, HW, prec, logits_out=out)
        # PREC bf16: exact products of the bf16 operands; split: drops lo*lo (2^-16 rel per product)
        tol = 1e-5 if prec == _lib.PH_PREC_BF16 else 5e-5
        assert Hh.rel_err(out.cpu().reshape(B, N, HW), ref) < tol
        out16 = torch.empty(B, N, H, W, device=gpu, dtype=torch.bfloat16)
        E.dynconv(xp, kern, kbias_d, br, N, HW, prec, logits_out=out16, out_dtype=_lib.PH_OUT_BF16)
        assert Hh.rel_err(out16.float().cpu().reshape(B, N, HW), ref) < 5e-3
        bits = torch.full((B, Npad, E.hw_padded(HW) // 32), -1, dtype=torch.int32, device=gpu)
        E.dynconv(xp, kern, kbias_d, br, N, HW, prec, bits_out=bits)
        got = unpack_bits(bits, N, HW)
        want = (out.cpu().reshape(B, N, HW).sigmoid() > 0.5).float()
        assert torch.equal(got, want)
        full = unpack_bits(bits, Npad, bits.shape[2] * 32)
        assert full[:, N:].sum() == 0 and full[:, :, HW:].sum() == 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("H,W", [(8, 16), (5, 7), (1, 3), (4, 256), (3, 512)])   # 256 / 512: whole waves per row (fp32 store exchange)
def test_upsample2x(gpu, dtype, H, W):
    x = torch.randn(3, 5, H, W, generator=torch.Generator().manual_seed(5)).to(dtype)
    out = E.upsample2x(x.to(gpu)).cpu().float()
    ref = F.interpolate(x.float(), scale_factor=2, mode="bilinear", align_corners=False)
    tol = 1e-6 if dtype == torch.float32 else 8e-3
    assert (out - ref).abs().max() <= tol * ref.abs().max()


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("H,W,B", [(8, 16, 2), (6, 13, 1)])
def test_khead_conv_gn_planes_entry_point(gpu, prec, H, W, B):
    """ph_khead_conv_gn, the pre-fusion entry point (loc / sem / x / depth planes out, loc read back through LDS for the
    sum): against torch conv1x1 -> GroupNorm(32) -> ReLU (kernel_head.py:250-251,264-265,277-278,303)"""
    g = torch.Generator().manual_seed(17)
    HW, P = H * W, (2 if prec == _lib.PH_PREC_SPLIT else 1)
    f = [torch.randn(B, 256, H, W, generator=g).relu() for _ in range(3)]
    w = torch.randn(3, 256, 256, generator=g) * 0.05
    gam, bet = 1 + 0.1 * torch.randn(3, 256, generator=g), 0.1 * torch.randn(3, 256, generator=g)
    ref = [F.relu(F.group_norm(F.conv2d(f[m].double(), w[m].double()[:, :, None, None]), 32, gam[m].double(), bet[m].double(), 1e-5))
           for m in range(3)]
    wh = w.to(torch.bfloat16)
    planes = [wh.view(torch.int16)]
    if P == 2:
        planes.append((w - wh.float()).to(torch.bfloat16).view(torch.int16))
    wpl = torch.stack(planes, 0).contiguous().to(gpu)                      # [P][3][256][256]
    gn = torch.stack([gam, bet], 1).contiguous().to(gpu)                   # [3][2][256]
    HWp = E.hw_padded(HW)
    mk = lambda: torch.full((P, B, 256, HWp), 0x7fc0, dtype=torch.int16, device=gpu)
    loc, sem, xs, dfe = mk(), mk(), mk(), mk()
    xf, df = torch.empty(B, 256, H, W, device=gpu), torch.empty(B, 256, H, W, device=gpu)
    lib = _lib.load()
    ws = torch.empty(lib.ph_khead_workspace_bytes(B, HW, 32), dtype=torch.uint8, device=gpu)
    fd = [t.to(gpu) for t in f]
    _lib.check(lib.ph_khead_conv_gn(_lib.ptr(fd[0]), _lib.ptr(fd[1]), _lib.ptr(fd[2]), _lib.ptr(wpl), _lib.ptr(gn), 32, 1e-5,
                                    _lib.ptr(loc), _lib.ptr(sem), _lib.ptr(xs), _lib.ptr(dfe), _lib.ptr(xf), _lib.ptr(df),
                                    _lib.ptr(ws), ws.numel(), B, HW, prec, _lib.stream_ptr()), "ph_khead_conv_gn")
    tol = 2e-2 if prec == _lib.PH_PREC_BF16 else 1e-4
    rec = lambda pl: planes_to_float(pl.cpu())[..., :HW].reshape(B, 256, H, W)
    for got, want in ((rec(loc), ref[0]), (rec(sem), ref[1]), (rec(dfe), ref[2]), (rec(xs), ref[0] + ref[1]),
                      (xf.cpu().double(), ref[0] + ref[1]), (df.cpu().double(), ref[2])):
        assert (got - want).abs().max() <= tol * want.abs().max()
    for pl in (loc, sem, xs, dfe):
        assert (planes_to_float(pl.cpu())[..., HW:] == 0).all()            # zero padding


def test_errors_are_reported(gpu):
    lib = _lib.load()
    assert lib.ph_version() == 100
    rc = lib.ph_pool(None, None, None, None, 1, 1, 128, 1, 1, None)
    assert rc == -1 and b"ph_pool" in lib.ph_last_error_string()
    with pytest.raises(_lib.PolyheadError):
        E.ingest(torch.zeros(1, 256, 4, 4), _lib.PH_PREC_BF16)       # CPU tensor: no fallback


# ---- round 2: fp16 planes, hi/lo kernels over one feature plane, fp16 outputs -------------------------------------------
def _planes16(t, dtype):
    return t.to(dtype).view(torch.int16)


@pytest.mark.parametrize("N,H,W,nsplit", [(153, 16, 32, 3), (40, 6, 13, 1), (253, 8, 48, 2)])
def test_pool_fp16_planes(gpu, N, H, W, nsplit):
    """PH_PREC_F16: {0, 1} x fp16 products are exact, fp32 accumulation"""
    g = torch.Generator().manual_seed(31)
    B, HW = 2, H * W
    x, d = torch.randn(B, 256, H, W, generator=g), torch.randn(B, 256, H, W, generator=g)
    m = torch.randn(B, N, H, W, generator=g)
    xp, dp = E.ingest(x.to(gpu), _lib.PH_PREC_F16), E.ingest(d.to(gpu), _lib.PH_PREC_F16)
    assert torch.equal(xp[0, :, :, :HW].reshape(B, 256, H, W).cpu(), _planes16(x, torch.float16))      # RNE fp16
    assert (xp[..., HW:] == 0).all()
    bits = E.binarize(m.to(gpu))
    got = E.pool(xp, dp, bits, N, HW, _lib.PH_PREC_F16, nsplit=nsplit).cpu().double().sum(1)[:, :N]
    M = (m.sigmoid() > 0.5).double().reshape(B, N, HW)
    xq, dq = x.half().double().reshape(B, 256, HW), d.half().double().reshape(B, 256, HW)
    ref = torch.cat([torch.einsum("bnk,bck->bnc", M, xq), torch.einsum("bnk,bck->bnc", M, dq)], -1)
    assert (got - ref).abs().max() <= 1e-5 * ref.abs().max()
    ref32 = torch.cat([torch.einsum("bnk,bck->bnc", M, x.double().reshape(B, 256, HW)),
                       torch.einsum("bnk,bck->bnc", M, d.double().reshape(B, 256, HW))], -1)
    assert Hh.rel_err(got, ref32) < 5e-4              # 2^-12 per element, 8x finer than a bf16 plane


@pytest.mark.parametrize("prec", [_lib.PH_PREC_BF16_KSPLIT, _lib.PH_PREC_F16, _lib.PH_PREC_BF16_KF16])
@pytest.mark.parametrize("N,H,W", [(111, 8, 16), (153, 16, 32), (40, 6, 13), (253, 8, 48), (20, 4, 32)])
def test_dynconv_ksplit_and_fp16(gpu, prec, N, H, W):
    """PH_PREC_BF16_KSPLIT: hi + lo kernel planes x ONE bf16 feature plane (2 MFMAs); PH_PREC_F16: fp16 x fp16;
    PH_PREC_BF16_KF16: ONE fp16 kernel plane x a bf16 feature plane whose fragments are converted to fp16 in registers
    (exact).  Products of the stored operands are exact in fp32 accumulation; outputs fp32 / fp16 / bits"""
    g = torch.Generator().manual_seed(41)
    B, HW = 2, H * W
    Npad = E.n_padded(N)
    x = torch.randn(B, 256, H, W, generator=g)
    kern_f = torch.randn(2, B, Npad, 256, generator=g) * 0.1
    kbias = torch.randn(2, B, Npad, generator=g) * 0.1
    if prec == _lib.PH_PREC_F16:
        kern = _planes16(kern_f, torch.float16)[None].contiguous().to(gpu)
        kq = kern_f.half().double()
        xp = E.ingest(x.to(gpu), _lib.PH_PREC_F16)
        xq = x.half().double().reshape(B, 256, HW)
    elif prec == _lib.PH_PREC_BF16_KF16:
        kern = _planes16(kern_f, torch.float16)[None].contiguous().to(gpu)
        kq = kern_f.half().double()
        xp = E.ingest(x.to(gpu), _lib.PH_PREC_BF16)
        xq = x.bfloat16().double().reshape(B, 256, HW)
    else:
        hi = kern_f.to(torch.bfloat16)
        lo = (kern_f - hi.float()).to(torch.bfloat16)
        kern = torch.stack([hi.view(torch.int16), lo.view(torch.int16)], 0).contiguous().to(gpu)
        kq = hi.double() + lo.double()
        xp = E.ingest(x.to(gpu), _lib.PH_PREC_BF16)
        xq = x.bfloat16().double().reshape(B, 256, HW)
    kbias_d = kbias.to(gpu)
    for br in (0, 1):
        ref = torch.einsum("bnc,bck->bnk", kq[br, :, :N], xq) + kbias[br, :, :N, None].double()
        out = torch.empty(B, N, H, W, device=gpu)
        E.dynconv(xp, kern, kbias_d, br, N, HW, prec, logits_out=out)
        assert Hh.rel_err(out.cpu().reshape(B, N, HW), ref) < 1e-5
        # and against the unrounded kernels: 2^-17 (hi/lo) / 2^-12 (fp16) per element
        ref_k = torch.einsum("bnc,bck->bnk", kern_f[br, :, :N].double(), xq) + kbias[br, :, :N, None].double()
        assert Hh.rel_err(out.cpu().reshape(B, N, HW), ref_k) < (2e-5 if prec == _lib.PH_PREC_BF16_KSPLIT else 5e-4)
        out16 = torch.empty(B, N, H, W, device=gpu, dtype=torch.float16)
        E.dynconv(xp, kern, kbias_d, br, N, HW, prec, logits_out=out16, out_dtype=_lib.PH_OUT_F16)
        assert Hh.rel_err(out16.float().cpu().reshape(B, N, HW), ref) < 2.0 ** -11           # one fp16 rounding
        assert torch.equal(out16.cpu(), out.cpu().half())                                    # exactly RNE of the fp32 result
        bits = torch.full((B, Npad, E.hw_padded(HW) // 32), -1, dtype=torch.int32, device=gpu)
        E.dynconv(xp, kern, kbias_d, br, N, HW, prec, bits_out=bits)
        assert torch.equal(unpack_bits(bits, N, HW), (out.cpu().reshape(B, N, HW).sigmoid() > 0.5).float())


@pytest.mark.parametrize("H,W", [(8, 16), (5, 7), (4, 256)])
def test_upsample2x_fp16(gpu, H, W):
    x = torch.randn(3, 5, H, W, generator=torch.Generator().manual_seed(6)).half()
    out = E.upsample2x(x.to(gpu))
    assert out.dtype == torch.float16
    ref = F.interpolate(x.float(), scale_factor=2, mode="bilinear", align_corners=False)
    assert (out.cpu().float() - ref).abs().max() <= 2.0 ** -10 * ref.abs().max()


def test_dynconv_kf16_register_conversion_form(gpu):
    """PH_CONV_COOP=0 (read once per process): the `mixed16` conv converting its bf16 fragments per wave in registers, the
    form every test ran until the cooperative in-LDS conversion became the default -- the same exactness cases in a child"""
    import os
    import subprocess
    import sys
    if os.environ.get("PH_CONV_COOP") == "0":
        pytest.skip("already the child process")
    env = dict(os.environ, PH_CONV_COOP="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-k", "test_dynconv_ksplit_and_fp16"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "15 passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("prec,dt", [(_lib.PH_PREC_BF16_KF16, torch.float16), (_lib.PH_PREC_F16, torch.float16), (_lib.PH_PREC_BF16, torch.bfloat16)])
@pytest.mark.parametrize("N,H,B,wgs", [(153, 6, 3, 0), (111, 16, 2, 5), (200, 3, 4, 3), (153, 128, 2, 0),
                                       # five row blocks (round 6: the last block's windows run on helper waves, one tile late): ranges of many
                                       # rows spanning frames, a last block with ONE live query-row group, with all 32 rows live, one-row frames
                                       (153, 16, 2, 5), (130, 7, 3, 3), (160, 5, 2, 2), (137, 1, 5, 2), (145, 2, 3, 0)])
def test_dynconv_up2_fused_final_stage(gpu, monkeypatch, prec, dt, N, H, B, wgs):
    """ph_dynconv_up2 = ph_dynconv (16-bit logits) + ph_upsample2x in one kernel (kernel_update_head.py:317-329 +
    kernel_update.py:131-143).  W = 256; workgroup ranges of one row (rows < CUs), of many rows spanning frames (PH_UP2_WGS),
    and cfg2's full map.  Low-resolution logits: bit-identical to the two-kernel form (same MFMA sequence).  Upsampled
    logits: within one 16-bit ulp of F.interpolate on those low-resolution values (fp32 blend, one final rounding), and almost
    everywhere bit-identical to ph_upsample2x (the blend is vertical-then-horizontal here, horizontal-then-vertical there)."""
    if wgs:
        monkeypatch.setenv("PH_UP2_WGS", str(wgs))
    W = 256
    lib = _lib.load()
    oc = E.OUT_CODE[dt]
    assert lib.ph_dynconv_up2_supported(N, H, W, prec, oc) == 1
    g = torch.Generator().manual_seed(43 + N + H)
    HW, Npad = H * W, E.n_padded(N)
    x = torch.randn(B, 256, H, W, generator=g)
    kern_f = torch.randn(2, B, Npad, 256, generator=g) * 0.1
    kbias = (torch.randn(2, B, Npad, generator=g) * 0.1).to(gpu)
    kdt = torch.bfloat16 if prec == _lib.PH_PREC_BF16 else torch.float16
    kern = _planes16(kern_f, kdt)[None].contiguous().to(gpu)
    xp = E.ingest(x.to(gpu), _lib.PH_PREC_F16 if prec == _lib.PH_PREC_F16 else _lib.PH_PREC_BF16)
    ulp = 2.0 ** -10 if dt == torch.float16 else 2.0 ** -7
    for br in (0, 1):
        low2 = torch.empty(B, N, H, W, device=gpu, dtype=dt)
        E.dynconv(xp, kern, kbias, br, N, HW, prec, logits_out=low2, out_dtype=oc)
        up2 = E.upsample2x(low2)
        low = torch.full((B, N, H, W), float("nan"), device=gpu, dtype=dt)
        up = torch.full((B, N, 2 * H, 2 * W), float("nan"), device=gpu, dtype=dt)
        E.dynconv_up2(xp, kern, kbias, br, N, H, W, prec, up, logits_out=low, out_dtype=oc)
        assert torch.equal(low, low2), br
        ref = F.interpolate(low2.float().cpu(), scale_factor=2, mode="bilinear", align_corners=False)
        d = (up.float().cpu() - ref).abs()
        assert not torch.isnan(up.float()).any()
        assert bool((d <= ulp * ref.abs() + 1e-7).all()), float((d / ref.abs().clamp_min(1e-6)).max())
        assert float((up != up2).float().mean()) < 2e-3
        # without the low-resolution output (the depth branch's form): the same upsampled tensor
        upn = torch.full_like(up, float("nan"))
        E.dynconv_up2(xp, kern, kbias, br, N, H, W, prec, upn, logits_out=None, out_dtype=oc)
        assert torch.equal(upn, up)


@pytest.mark.parametrize("prec,feat,kdt", [(_lib.PH_PREC_BF16_KF16, _lib.PH_PREC_BF16, torch.float16), (_lib.PH_PREC_F16, _lib.PH_PREC_F16, torch.float16),
                                           (_lib.PH_PREC_BF16, _lib.PH_PREC_BF16, torch.bfloat16)])
@pytest.mark.parametrize("N,H,W,B,ns", [(153, 128, 256, 2, 10), (111, 16, 24, 3, 3), (160, 9, 31, 2, 4), (40, 7, 64, 1, 2), (192, 6, 20, 2, 1), (153, 48, 156, 2, 7)])
def test_dynconv_poolx_equals_dynconv_and_pool(gpu, prec, feat, kdt, N, H, W, B, ns):
    """round 6: ph_dynconv_poolx = ph_dynconv (mask bits, bit for bit) + the x half of ph_pool with THOSE bits (kernel_update_head.py
    :317-329, :236-241) from one read of the plane; the depth half by ph_pool_counts on the depth plane alone (columns 256 .. 511, bit
    for bit the full pooling's at the same split).  Ragged sizes: H * W not a multiple of 64 / 128, N not a multiple of 32, ranges of
    uneven tile counts, one range.  The pooled x sums: exact {0, 1} x plane products in both kernels, different split boundaries
    and, in the KF16 grade, the tile converted to fp16 first -- compared after the fixed-order sum over the ranges at 1e-6"""
    lib = _lib.load()
    assert lib.ph_dynconv_poolx_supported(N, prec) == 1
    g = torch.Generator().manual_seed(7 + N + H)
    HW, Npad = H * W, E.n_padded(N)
    x, d = torch.randn(B, 256, H, W, generator=g), torch.randn(B, 256, H, W, generator=g)
    kern = _planes16(torch.randn(2, B, Npad, 256, generator=g) * 0.1, kdt)[None].contiguous().to(gpu)
    kbias = (torch.randn(2, B, Npad, generator=g) * 0.1).to(gpu)
    xp, dp = E.ingest(x.to(gpu), feat), E.ingest(d.to(gpu), feat)
    HWp = E.hw_padded(HW)
    bits_ref = torch.full((B, Npad, HWp // 32), -1, dtype=torch.int32, device=gpu)
    E.dynconv(xp, kern, kbias, 0, N, HW, prec, bits_out=bits_ref)
    part_ref = torch.zeros((B, ns, Npad, 512), dtype=torch.float32, device=gpu)
    cnt_ref = torch.zeros((B, ns, Npad), dtype=torch.int32, device=gpu)
    E.pool(xp, dp, bits_ref, N, HW, feat, ns, out=part_ref, counts=cnt_ref)
    bits = torch.full_like(bits_ref, -1)
    part = torch.full((B, ns, Npad, 512), float("nan"), dtype=torch.float32, device=gpu)
    cnt = torch.full((B, ns, Npad), -1, dtype=torch.int32, device=gpu)
    E.dynconv_poolx(xp, kern, kbias, N, HW, prec, bits, part)
    assert torch.equal(bits[:, :N], bits_ref[:, :N])
    E.pool_depth_only(dp, bits, N, HW, feat, part, cnt)
    assert not torch.isnan(part).any()
    assert torch.equal(part[..., 256:], part_ref[..., 256:]) and torch.equal(cnt.sum(1), cnt_ref.sum(1))
    a, b = part[..., :256].double().sum(1).cpu(), part_ref[..., :256].double().sum(1).cpu()
    assert Hh.rel_err(a, b) < 1e-6, Hh.rel_err(a, b)
    # the kernel's pixel ranges are k_pool's: in the grades that pool the plane as it is (bf16, fp16) every range's sums are k_pool's BIT
    # FOR BIT -- what lets batch-invariant plans choose between the two forms by launch size (engine.DecodePlan)
    if prec != _lib.PH_PREC_BF16_KF16:
        assert torch.equal(part[:, :, :N, :256], part_ref[:, :, :N, :256])


def test_dynconv_up2_inside_the_decode_plan(gpu, monkeypatch):
    """the S-stage plan with and without the fused final stage (PH_CONV_UP2=0): every output the API returns agrees -- the
    low-resolution mask logits and the query outputs bit for bit, the upsampled logits except for rounding-boundary cases"""
    import bench
    wl = dict(H=8, W=256, Nq=100, n_thing=8, n_stuff=11, S=2, F=2048)
    N = wl["Nq"] + wl["n_stuff"]
    inp = bench.synth_inputs(wl, 3, seed=8)
    outs = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("PH_CONV_UP2", fused)
        head = bench.build_head(wl, "mixed16", torch.float16, gpu, seed=4)
        plan = head._plan(3, N, wl["H"], wl["W"], gpu)
        assert plan.fused_up == (fused == "1")
        plan.set_inputs(*[inp[k].to(gpu) for k in ("x", "dfe", "k0", "q0", "m0")])
        plan.run()
        torch.cuda.synchronize()
        outs[fused] = {k: (None if v is None else v.clone()) for k, v in plan.outputs().items()}
    assert outs["1"]["depth"] is None and outs["0"]["depth"] is not None
    for k in ("obj", "dobj", "cls", "mask"):
        assert torch.equal(outs["1"][k], outs["0"][k]), k
    for k in ("mask_up", "depth_up"):
        assert float((outs["1"][k] != outs["0"][k]).float().mean()) < 2e-3, k
        assert Hh.rel_err(outs["1"][k].float().cpu(), outs["0"][k].float().cpu()) < 2e-3, k


@pytest.mark.parametrize("precision", ["mixed16", "fp16"])
def test_dynconv_poolx_inside_the_decode_plan(gpu, monkeypatch, precision):
    """the S-stage plan at cfg2's map size with and without the fused conv + pooling (PH_CONV_POOLX=0), 8 frames, 3 stages: the pooled
    x sums of stages 1, 2 differ by fp32 summation order (another pixel split) -- the first stage's outputs, which do not depend on
    them, are bit-identical; everything after agrees at the grade's own rounding level (a mask bit next to the threshold may flip)"""
    import bench
    wl = dict(H=128, W=256, Nq=100, n_thing=8, n_stuff=11, S=3, F=2048)
    N = wl["Nq"] + wl["n_stuff"]
    B = 8
    inp = bench.synth_inputs(wl, B, seed=8)
    outs, stage0, stage1 = {}, {}, {}
    for fused in ("1", "0"):
        monkeypatch.setenv("PH_CONV_POOLX", fused)
        head = bench.build_head(wl, precision, torch.float16, gpu, seed=4)
        plan = head._plan(B, N, wl["H"], wl["W"], gpu)
        assert plan.poolx == (fused == "1")
        plan.set_inputs(*[inp[k].to(gpu) for k in ("x", "dfe", "k0", "q0", "m0")])
        plan.run()
        torch.cuda.synchronize()
        outs[fused] = {k: (None if v is None else v.clone()) for k, v in plan.outputs().items()}
        stage0[fused] = {k: v.clone() for k, v in plan.stage_out[0].items()}
        stage1[fused] = {k: v.clone() for k, v in plan.stage_out[1].items()}
    for k in stage0["1"]:
        assert torch.equal(stage0["1"][k], stage0["0"][k]), k
    # stage 1 has seen ONE fused boundary: its pooled x sums differ in the last fp32 bits, which may move a 16-bit rounding of a dynamic
    # kernel and with it a mask bit next to the threshold; the last stage has seen two and the pooling of such bits.  The bounds are the
    # 16-bit grades' own (tests/test_gpu_fullsize.py: 3e-2 against the oracle); a wiring error (wrong buffer, wrong counts) is O(1)
    errs = {k: Hh.rel_err(stage1["1"][k].float().cpu(), stage1["0"][k].float().cpu()) for k in ("obj", "dobj", "cls", "kbias")}
    errs.update({"final_" + k: Hh.rel_err(outs["1"][k].float().cpu(), outs["0"][k].float().cpu()) for k in ("obj", "dobj", "cls", "mask_up", "depth_up")})
    flips = {k: float(((outs["1"][k] > 0) != (outs["0"][k] > 0)).float().mean()) for k in ("mask_up",)}
    print(errs, flips)
    assert all(e < 5e-3 for k, e in errs.items() if not k.startswith("final_")), errs
    assert all(e < 3e-2 for e in errs.values()), errs
    assert flips["mask_up"] < 5e-3, flips
