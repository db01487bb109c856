"""SURVEY.md 8(f) N3 on the device: channels-last conv / GroupNorm kernels against torch, and SemanticFPNWrapper
against the reference's golden outputs and the oracle."""
import json
import os

import pytest
import torch
import torch.nn.functional as F

import helpers as Hh
from oracle import neck_oracle as NO
from polyphonicformer_amd import _lib, engine as E
from polyphonicformer_amd.pack import pack_b32
from polyphonicformer_amd.registry import NECKS
import polyphonicformer_amd.semantic_fpn as SF  # noqa: F401

pytestmark = pytest.mark.gpu


def _planes_to_float(p):
    """int16 bf16 planes [P, ...] -> float64 sum of planes"""
    return sum((p[i].to(torch.int32) << 16).view(torch.float32).double() for i in range(p.shape[0]))


@pytest.mark.parametrize("prec", [_lib.PH_PREC_BF16, _lib.PH_PREC_SPLIT])
@pytest.mark.parametrize("k,s,H,W,B", [(3, 1, 6, 70, 2), (3, 2, 9, 131, 2), (1, 1, 5, 64, 2), (3, 1, 2, 3, 2), (3, 1, 64, 1024, 2),
                                       (3, 2, 72, 2048, 2), (3, 1, 6, 70, 1), (3, 1, 32, 64, 1), (3, 1, 48, 512, 5)])
def test_conv_nhwc(gpu, prec, k, s, H, W, B):
    # the small maps take the 2-row tiles (fewer than 256 four-row tiles per launch), the large ones the 4-row tiles (bf16 grade);
    # B = 1 on a small map and B = 5 (where the choice follows the LAUNCH's tile count): the partial-sum layout ph_gn_finalize
    # reads is the one ph_conv_nhwc_workgroups_b(..., B) describes (the B-less query is removed)
    g = torch.Generator().manual_seed(7)
    P = 2 if prec == _lib.PH_PREC_SPLIT else 1
    x = torch.randn(B, 256, H, W, generator=g)
    w = torch.randn(256, 256, k, k, generator=g) * 0.05
    xp = torch.empty((P, B, H * W, 256), dtype=torch.int16, device=gpu)
    E.nhwc_ingest(x.to(gpu), None, prec, xp)
    xq = _planes_to_float(xp.cpu()).reshape(B, H, W, 256).permute(0, 3, 1, 2)             # what the kernel sees
    w2 = w.double().permute(0, 2, 3, 1).reshape(256, -1)
    wpl = E._planes_of(w2, P)
    wq = _planes_to_float(wpl).reshape(256, k, k, 256).permute(0, 3, 1, 2)
    wp = torch.stack([pack_b32(wpl[p]) for p in range(P)], 0).contiguous().to(gpu)
    ref = F.conv2d(xq, wq, None, stride=s, padding=k // 2)                                  # float64
    Ho, Wo = ref.shape[-2:]
    y = torch.empty((B, Ho * Wo, 256), dtype=torch.float32, device=gpu)
    lib = _lib.load()
    partial = torch.zeros((lib.ph_conv_nhwc_partial_floats(B, Ho, Wo),), dtype=torch.float32, device=gpu)
    E.conv_nhwc(xp, dict(wp=wp, k=k, s=s), y, partial, B, H, W, prec)
    got = y.cpu().reshape(B, Ho, Wo, 256).permute(0, 3, 1, 2).double()
    tol = 1e-5 if prec == _lib.PH_PREC_BF16 else 5e-5       # exact products of the bf16 operands / dropped lo*lo terms
    assert Hh.rel_err(got, ref) < tol
    nwg = lib.ph_conv_nhwc_workgroups_b(k, s, Ho, Wo, prec, B)
    pr = partial.cpu()[:B * nwg * 512].reshape(B, nwg, 256, 2).double().sum(1)   # the buffer is sized for the upper bound
    assert Hh.rel_err(pr[..., 0], got.sum((2, 3))) < 1e-4 and Hh.rel_err(pr[..., 1], (got * got).sum((2, 3))) < 1e-4
    stats = torch.empty((B, 32, 2), dtype=torch.float32, device=gpu)
    E.gn_finalize(partial, stats, nwg, 32, Ho * Wo, B)
    gm = got.reshape(B, 32, -1)
    assert Hh.rel_err(stats.cpu()[..., 0], gm.mean(2)) < 1e-4
    assert Hh.rel_err(stats.cpu()[..., 1], 1.0 / torch.sqrt(gm.var(2, unbiased=False) + 1e-5)) < 1e-4


@pytest.mark.parametrize("prec", [_lib.PH_PREC_BF16, _lib.PH_PREC_F16])
@pytest.mark.parametrize("k,s,H,W,B", [(3, 1, 6, 70, 2), (3, 1, 33, 130, 1), (3, 2, 19, 131, 2), (1, 1, 7, 64, 3), (3, 1, 128, 256, 1)])
def test_conv_tile_forms_give_identical_bits(gpu, monkeypatch, prec, k, s, H, W, B):
    """round 6: the 2-row and the 4-row tile forms of k_conv_nhwc write the same conv output AND the same GroupNorm
    partial sums -- one entry per pair of output rows, in the 2-row kernel's order -- so the form a launch's size picks never shows in
    a frame's bits (odd row counts: the last pair is one row; the last 4-row tile may hold one pair only)"""
    g = torch.Generator().manual_seed(11 + H + W)
    x = torch.randn(B, 256, H, W, generator=g).to(gpu)
    w = torch.randn(256, 256, k, k, generator=g) * 0.05
    wpl = E._planes_of(w.double().permute(0, 2, 3, 1).reshape(256, -1), 1, prec == _lib.PH_PREC_F16)
    wp = pack_b32(wpl[0])[None].contiguous().to(gpu)
    xp = torch.empty((1, B, H * W, 256), dtype=torch.int16, device=gpu)
    E.nhwc_ingest(x, None, prec, xp)
    Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
    lib = _lib.load()
    nwg = lib.ph_conv_nhwc_workgroups_b(k, s, Ho, Wo, prec, B)
    assert nwg == ((Wo + 63) // 64) * ((Ho + 1) // 2)
    outs = []
    for th in ("2", "4"):
        monkeypatch.setenv("PH_CONV_TH_NOW", th)
        y = torch.empty((B, Ho * Wo, 256), dtype=torch.float32, device=gpu)
        partial = torch.full((lib.ph_conv_nhwc_partial_floats(B, Ho, Wo),), float("nan"), dtype=torch.float32, device=gpu)
        E.conv_nhwc(xp, dict(wp=wp, k=k, s=s), y, partial, B, H, W, prec)
        stats = torch.empty((B, 32, 2), dtype=torch.float32, device=gpu)
        E.gn_finalize(partial, stats, nwg, 32, Ho * Wo, B)
        outs.append((y.cpu(), partial.cpu()[:B * nwg * 512], stats.cpu()))
    monkeypatch.delenv("PH_CONV_TH_NOW")
    assert not torch.isnan(outs[0][1]).any()                      # every entry the finalize pass reads is written
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("prec", [_lib.PH_PREC_BF16, _lib.PH_PREC_F16])
@pytest.mark.parametrize("H,W,B", [(9, 131, 2), (72, 2048, 2), (256, 512, 1), (7, 5, 3)])
def test_chunk_major_planes_for_the_stride2_conv(gpu, prec, H, W, B):
    """PH_PLANES_C16: ph_nhwc_ingest writes [B][16][HW][16], the 3x3 stride-2 kernel reads it -- every plane element is the
    channels-last plane's, and the conv output and GroupNorm partial sums keep their bits (same products, same order)"""
    g = torch.Generator().manual_seed(H + W)
    x = (torch.randn(B, 256, H, W, generator=g) * 2).to(gpu)
    w = torch.randn(256, 256, 3, 3, generator=g) * 0.05
    wpl = E._planes_of(w.double().permute(0, 2, 3, 1).reshape(256, -1), 1, prec == _lib.PH_PREC_F16)
    wp = pack_b32(wpl[0])[None].contiguous().to(gpu)
    nhwc = torch.empty((1, B, H * W, 256), dtype=torch.int16, device=gpu)
    c16 = torch.full((1, B, H * W, 256), 0x7FFF, dtype=torch.int16, device=gpu)
    E.nhwc_ingest(x, None, prec, nhwc)
    E.nhwc_ingest(x, None, prec | _lib.PH_PLANES_C16, c16)
    assert torch.equal(c16.view(B, 16, H * W, 16).permute(0, 2, 1, 3).reshape(B, H * W, 256), nhwc[0])
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    lib = _lib.load()
    outs = []
    for planes, flag in ((nhwc, 0), (c16, _lib.PH_PLANES_C16)):
        y = torch.empty((B, Ho * Wo, 256), dtype=torch.float32, device=gpu)
        partial = torch.zeros((lib.ph_conv_nhwc_partial_floats(B, Ho, Wo),), dtype=torch.float32, device=gpu)
        E.conv_nhwc(planes, dict(wp=wp, k=3, s=2), y, partial, B, H, W, prec | flag)
        outs.append((y.cpu(), partial.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert lib.ph_conv_nhwc_workgroups_b(3, 2, Ho, Wo, prec | _lib.PH_PLANES_C16, B) == lib.ph_conv_nhwc_workgroups_b(3, 2, Ho, Wo, prec, B)
    # refused where it has no meaning: stride 1, the two-plane format
    with pytest.raises(_lib.PolyheadError):
        E.conv_nhwc(c16, dict(wp=wp, k=3, s=1), y, partial, B, H, W, prec | _lib.PH_PLANES_C16)
    with pytest.raises(_lib.PolyheadError):
        E.nhwc_ingest(x, None, _lib.PH_PREC_SPLIT | _lib.PH_PLANES_C16, torch.empty((2, B, H * W, 256), dtype=torch.int16, device=gpu))


@pytest.mark.parametrize("prec", [_lib.PH_PREC_BF16, _lib.PH_PREC_F16, _lib.PH_PREC_SPLIT])
@pytest.mark.parametrize("H,W,with_add", [(128, 256, False), (16, 32, True), (6, 70, False), (3, 28, True), (5, 7, False)])
def test_nhwc_ingest_exact(gpu, prec, H, W, with_add):
    """fp32 NCHW (+ the positional encoding of level 3) -> 16-bit NHWC planes: every element is the correctly rounded
    input, in the 16-byte form (H*W % 4 == 0) and the scalar form, ragged last tile included"""
    g = torch.Generator().manual_seed(H * W)
    B = 2
    x = torch.randn(B, 256, H, W, generator=g) * 3
    add = torch.randn(256, H, W, generator=g) if with_add else None
    P = 2 if prec == _lib.PH_PREC_SPLIT else 1
    out = torch.full((P, B, H * W, 256), 0x7FFF, dtype=torch.int16, device=gpu)
    E.nhwc_ingest(x.to(gpu), None if add is None else add.to(gpu), prec, out)
    v = (x + add[None]) if with_add else x
    v = v.permute(0, 2, 3, 1).reshape(B, H * W, 256)
    got = out.cpu()
    if prec == _lib.PH_PREC_F16:
        assert torch.equal(got[0].view(torch.float16), v.half())
    else:
        hi = v.bfloat16()
        assert torch.equal(got[0].view(torch.bfloat16), hi)
        if P == 2:
            assert torch.equal(got[1].view(torch.bfloat16), (v - hi.float()).bfloat16())


@pytest.mark.parametrize("H,W", [(5, 7), (8, 16)])
def test_gn_apply_modes(gpu, H, W):
    g = torch.Generator().manual_seed(9)
    B, G = 2, 32
    y = torch.randn(B, 256, H, W, generator=g) * 2 + 0.3
    gamma, beta = 1 + 0.1 * torch.randn(256, generator=g), 0.1 * torch.randn(256, generator=g)
    ycl = y.permute(0, 2, 3, 1).contiguous().to(gpu)
    gm = y.reshape(B, G, -1)
    stats = torch.stack([gm.mean(2), 1.0 / torch.sqrt(gm.var(2, unbiased=False) + 1e-5)], 2).contiguous().to(gpu)
    pk = dict(gamma=gamma.to(gpu), beta=beta.to(gpu))
    ref = F.group_norm(y, G, gamma, beta, 1e-5).relu()
    prec = _lib.PH_PREC_SPLIT
    pl = torch.empty((2, B, H * W, 256), dtype=torch.int16, device=gpu)
    E.gn_apply(ycl, stats, pk, G, _lib.PH_GN_TO_PLANES, B, H, W, prec, planes=pl)
    got = _planes_to_float(pl.cpu()).reshape(B, H, W, 256).permute(0, 3, 1, 2)
    assert Hh.rel_err(got, ref) < 2e-5
    up = torch.empty((2, B, 4 * H * W, 256), dtype=torch.int16, device=gpu)
    E.gn_apply(ycl, stats, pk, G, _lib.PH_GN_UP2_PLANES, B, H, W, prec, planes=up)
    got = _planes_to_float(up.cpu()).reshape(B, 2 * H, 2 * W, 256).permute(0, 3, 1, 2)
    assert Hh.rel_err(got, F.interpolate(ref, scale_factor=2, mode="bilinear", align_corners=False)) < 2e-5
    acc = torch.full((B, H * W, 256), 1.5, dtype=torch.float32, device=gpu)
    E.gn_apply(ycl, stats, pk, G, _lib.PH_GN_ACCUM, B, H, W, prec, outf=acc, accumulate=True)
    assert Hh.rel_err(acc.cpu().reshape(B, H, W, 256).permute(0, 3, 1, 2), ref + 1.5) < 1e-6
    E.gn_apply(ycl, stats, pk, G, _lib.PH_GN_ACCUM, B, H, W, prec, outf=acc, accumulate=False)
    assert Hh.rel_err(acc.cpu().reshape(B, H, W, 256).permute(0, 3, 1, 2), ref) < 1e-6
    nchw = torch.empty((B, 256, H, W), dtype=torch.float32, device=gpu)
    E.gn_apply(ycl, stats, pk, G, _lib.PH_GN_TO_NCHW, B, H, W, prec, outf=nchw)
    assert Hh.rel_err(nchw.cpu(), ref) < 1e-6
    E.gn_apply(ycl, None, None, G, _lib.PH_GN_TO_PLANES, B, H, W, prec, planes=pl)        # plain conversion
    assert Hh.rel_err(_planes_to_float(pl.cpu()).reshape(B, H, W, 256).permute(0, 3, 1, 2), y) < 2e-5


def _neck(precision, gpu, sd):
    cfg = dict(type="SemanticFPNWrapper", in_channels=256, feat_channels=256, out_channels=256, start_level=0, end_level=3,
               upsample_times=2, positional_encoding=dict(type="SinePositionalEncoding", num_feats=128, normalize=True),
               cat_coors=False, cat_coors_level=3, fuse_by_cat=False, return_list=False, num_aux_convs=2,
               norm_cfg=dict(type="GN", num_groups=32, requires_grad=True))
    m = NECKS.build(cfg)
    m.load_state_dict(sd)
    m.eval().to(gpu)
    m.set_precision(precision)
    return m


def _state():
    keys = json.load(open(os.path.join(Hh.GOLDEN, "neck_state_keys.json")))["full"]
    return Hh.seeded_fill({k: tuple(v) for k, v in keys.items()}, 31)


@pytest.mark.parametrize("prec", [_lib.PH_PREC_BF16, _lib.PH_PREC_F16, _lib.PH_PREC_SPLIT])
@pytest.mark.parametrize("channels_last", [True, False])
@pytest.mark.parametrize("H,W,B", [(5, 7, 2), (8, 16, 2), (3, 70, 2), (9, 200, 5)])
def test_neck_out_convs(gpu, prec, channels_last, H, W, B):
    """ph_neck_out_convs (round 4): conv_pred + 2 aux convs = 1x1 conv + GroupNorm(32) + ReLU of one level sum, statistics from a
    recompute pass -- against torch on the 16-bit operands the kernel sees; plane and fp32 NCHW outputs; both input layouts"""
    g = torch.Generator().manual_seed(11)
    HW = H * W                 # B = 5: more than 3 frames (tile runs sized by the batch), 29 tiles per frame, a ragged last tile
    P, f16 = (2 if prec == _lib.PH_PREC_SPLIT else 1), prec == _lib.PH_PREC_F16
    dec = (lambda p: p[0].view(torch.float16).double()) if f16 else _planes_to_float
    s = F.relu(torch.randn(B, 256, H, W, generator=g)) * 2.0                                   # a level sum is >= 0
    w = torch.randn(3, 256, 256, generator=g) * 0.05
    gn = torch.stack([1.0 + 0.2 * torch.randn(3, 256, generator=g), 0.1 * torch.randn(3, 256, generator=g)], 1).contiguous()   # [3,2,256]
    wpl = E._planes_of(w.double(), P, f16)
    wq = dec(wpl)
    sp_cl = E._planes_of(s.double().permute(0, 2, 3, 1).reshape(B, HW, 256), P, f16)           # [P,B,HW,256]
    sq = dec(sp_cl).reshape(B, H, W, 256).permute(0, 3, 1, 2)
    HWp = E.hw_padded(HW)
    if channels_last:
        sp = sp_cl.to(gpu)
    else:
        sp = torch.zeros((P, B, 256, HWp), dtype=torch.int16)
        sp[..., :HW] = sp_cl.reshape(P, B, HW, 256).permute(0, 1, 3, 2)
        sp = sp.to(gpu)
    lib = _lib.load()
    ws = torch.empty((lib.ph_neck_out_convs_workspace_bytes(B, HW, 32) // 4 + 64,), dtype=torch.float32, device=gpu)
    outp = [torch.full((P, B, 256, HWp), 0x7F7F, dtype=torch.int16, device=gpu) for _ in range(3)]
    outf = [torch.full((B, 256, H, W), float("nan"), device=gpu) for _ in range(3)]
    # maps 0 and 2: both output kinds at once; map 1: planes only in one call, fp32 only in the other
    E.neck_out_convs(sp, channels_last, wpl.to(gpu), gn.to(gpu), 32, outp, [outf[0], None, outf[2]], ws, B, HW, prec)
    E.neck_out_convs(sp, channels_last, wpl.to(gpu), gn.to(gpu), 32, [outp[0], None, outp[2]], outf, ws, B, HW, prec)
    tol = 2e-5 if prec == _lib.PH_PREC_SPLIT else 1e-4       # fp32 accumulation of exact products (+ dropped lo*lo terms)
    otol = {_lib.PH_PREC_SPLIT: 2e-5, _lib.PH_PREC_BF16: 4e-3, _lib.PH_PREC_F16: 5e-4}[prec]   # rounding of the plane outputs
    for m in range(3):
        y = F.conv2d(sq, wq[m].reshape(256, 256, 1, 1))
        ref = F.relu(F.group_norm(y, 32, gn[m, 0].double(), gn[m, 1].double(), 1e-5))
        assert Hh.rel_err(outf[m].cpu().double(), ref) < tol, m
        pl = outp[m].cpu()
        assert Hh.rel_err(dec(pl)[..., :HW].reshape(B, 256, H, W), ref) < otol, m
        assert int(pl[..., HW:].abs().max()) == 0 if HWp > HW else True                        # planes are zero padded


@pytest.mark.parametrize("B", [3, 8])
def test_neck_output_stage_forms_agree_at_full_size(gpu, B):
    """cfg2's FPN sizes, fp16 grade, 3 frames (one stream, frame-sized tile runs) and 8 frames (the four tower streams, batch-sized
    tile runs): the recompute output stage (ph_neck_out_convs) against the per-map conv -> finalize -> apply form on the same
    plan -- fp32 NCHW outputs to summation order, plane outputs to one rounding of the 16-bit format"""
    torch.manual_seed(2)
    m = NECKS.build(dict(type="SemanticFPNWrapper", in_channels=256, feat_channels=256, out_channels=256, start_level=0, end_level=3,
                         upsample_times=2, positional_encoding=dict(type="SinePositionalEncoding", num_feats=128, normalize=True),
                         cat_coors=False, cat_coors_level=3, fuse_by_cat=False, return_list=False, num_aux_convs=2,
                         norm_cfg=dict(type="GN", num_groups=32, requires_grad=True)))
    m.init_weights(); m.eval().to(gpu); m.set_precision("fp16")
    g = torch.Generator().manual_seed(4)
    feats = [torch.randn(B, 256, 256 >> i, 512 >> i, generator=g).to(gpu) for i in range(4)]
    for planes in (False, True):
        run = (lambda: m.forward_planes(feats)) if planes else (lambda: m(feats))
        new = [o.clone() for o in run()]
        plan = next(iter(m._plans.values()))
        assert plan.out2 and plan.multi == (B >= 4)
        plan.out2 = False
        old = [o.clone() for o in run()]
        plan.out2 = True
        dec = (lambda t: t[0].view(torch.float16).float()) if planes else (lambda t: t)
        for a, b in zip(new, old):
            assert Hh.rel_err(dec(a).cpu(), dec(b).cpu()) < (1e-3 if planes else 1e-6)


@pytest.mark.parametrize("precision", ["fp32", "bf16", "fp16"])
def test_neck_vs_reference_golden(gpu, precision):
    sd = _state()
    m = _neck(precision, gpu, sd)
    feats = Hh.fpn_inputs(seed=32, B=1, C=256, H0=16, W0=32)
    outs = m([f.to(gpu) for f in feats])
    g = Hh.load_golden("full_neck.npz")
    tol = 3e-2 if precision == "bf16" else 1e-3          # 'fp16': one fp16 plane of weights / activations, f16 MFMA
    for name, o in zip(("out", "aux0", "aux1"), outs):
        e = Hh.rel_err(o.cpu(), torch.from_numpy(g[name]))
        print("neck", precision, name, e)
        assert e < tol, (name, e)


@pytest.mark.parametrize("precision", ["fp32", "fp16"])
def test_neck_vs_oracle_ragged_sizes(gpu, precision):
    """level sizes that are multiples of nothing (24x40 -> 12x20 -> 6x10 -> 3x5), two frames"""
    sd = _state()
    m = _neck(precision, gpu, sd)
    feats = Hh.fpn_inputs(seed=33, B=2, C=256, H0=24, W0=40)
    outs = m([f.to(gpu) for f in feats])
    ref = NO.semantic_fpn(sd, feats, groups=32, num_feats=128)
    for o, r in zip(outs, ref):
        assert tuple(o.shape) == tuple(r.shape) == (2, 256, 12, 20)
        assert Hh.rel_err(o.cpu(), r) < 1e-3


def test_kernel_head_with_its_neck(gpu):
    """KernelHead built from the shipped config (localization_fpn = SemanticFPNWrapper) takes the FPN tuple, as
    Polyphonic.simple_test hands it over (polyphonic_former.py:146-148): FPN levels -> neck -> post-neck, against
    the two oracles chained"""
    from oracle import poly_oracle as O
    from polyphonicformer_amd.registry import HEADS
    import polyphonicformer_amd.kernel_head  # noqa: F401
    neck_cfg = dict(type="SemanticFPNWrapper", in_channels=256, feat_channels=256, out_channels=256, start_level=0, end_level=3,
                    upsample_times=2, positional_encoding=dict(type="SinePositionalEncoding", num_feats=128, normalize=True),
                    cat_coors=False, cat_coors_level=3, fuse_by_cat=False, return_list=False, num_aux_convs=2,
                    norm_cfg=dict(type="GN", num_groups=32, requires_grad=True))
    kh = HEADS.build(dict(type="KernelHead", num_proposals=100, num_classes=19, num_thing_classes=8, num_stuff_classes=11,
                          in_channels=256, out_channels=256, cat_stuff_mask=True, feat_downsample_stride=2, feat_refine=False,
                          use_binary=True, conv_normal_init=True, proposal_feats_with_obj=True, kernel_init_std=1,
                          loss_seg=dict(type="FocalLoss", use_sigmoid=True), localization_fpn=neck_cfg))
    sd = Hh.seeded_fill({k: tuple(v.shape) for k, v in kh.state_dict().items()}, 41)
    kh.load_state_dict(sd)
    kh.eval().to(gpu)
    kh.set_precision("fp32")
    feats = Hh.fpn_inputs(seed=42, B=1, C=256, H0=16, W0=24)
    out = kh.simple_test_rpn(tuple(f.to(gpu) for f in feats), [Hh.img_meta(64, 96)])
    nsd = {k[len("localization_fpn."):]: v for k, v in sd.items() if k.startswith("localization_fpn.")}
    hsd = {k: v for k, v in sd.items() if not k.startswith("localization_fpn.")}
    maps = NO.semantic_fpn(nsd, feats, groups=32, num_feats=128)
    ref = O.kernel_head_post_neck(hsd, *maps, 8, 19, 32)
    for name, t in (("x_feats", out[1]), ("seg_preds", out[4]), ("depth_feats", out[5]), ("depth_pred", out[7])):
        assert Hh.rel_err(t.cpu(), ref[name]) < 1e-3, name
    assert Hh.rel_err(out[2].cpu(), ref["mask_preds"]) < 1e-3


@pytest.mark.parametrize("precision", ["bf16", "fp32", "fp16"])
def test_plane_handoff_equals_fp32_boundary(gpu, precision):
    """KernelHead with its neck takes the neck's maps as bf16 planes (SemanticFPNWrapper.forward_planes ->
    ph_khead_fused PH_IN_PLANES); the same head fed the neck's fp32 NCHW maps through the reference boundary must give
    the same result: bit-identical in bf16 precision (the first use of an fp32 map is the same rounding), to hi/lo
    accuracy in fp32 precision."""
    from polyphonicformer_amd.registry import HEADS
    import polyphonicformer_amd.kernel_head  # noqa: F401
    neck_cfg = dict(type="SemanticFPNWrapper", in_channels=256, feat_channels=256, out_channels=256, start_level=0, end_level=3,
                    upsample_times=2, positional_encoding=dict(type="SinePositionalEncoding", num_feats=128, normalize=True),
                    cat_coors=False, cat_coors_level=3, fuse_by_cat=False, return_list=False, num_aux_convs=2,
                    norm_cfg=dict(type="GN", num_groups=32, requires_grad=True))
    cfg = dict(type="KernelHead", num_proposals=100, num_classes=19, num_thing_classes=8, num_stuff_classes=11,
               in_channels=256, out_channels=256, cat_stuff_mask=True, feat_downsample_stride=2, feat_refine=False,
               use_binary=True, conv_normal_init=True, proposal_feats_with_obj=True, kernel_init_std=1,
               loss_seg=dict(type="FocalLoss", use_sigmoid=True))
    kh = HEADS.build(dict(cfg, localization_fpn=neck_cfg))
    sd = Hh.seeded_fill({k: tuple(v.shape) for k, v in kh.state_dict().items()}, 43)
    kh.load_state_dict(sd)
    kh.eval().to(gpu)
    kh.set_precision(precision)
    bare = HEADS.build(dict(cfg, localization_fpn=None))
    bare.load_state_dict({k: v for k, v in sd.items() if not k.startswith("localization_fpn.")})
    bare.eval().to(gpu)
    bare.set_precision(precision)
    feats = tuple(f.to(gpu) for f in Hh.fpn_inputs(seed=44, B=2, C=256, H0=24, W0=40))     # stride-8 map 12 x 20 (HW % 128 != 0)
    metas = [Hh.img_meta(96, 160)] * 2
    a = kh.simple_test_rpn(feats, metas)
    maps = [m.clone() for m in kh.localization_fpn(feats)]
    b = bare.simple_test_rpn(maps, metas)
    for i, name in ((0, "proposal_feats"), (1, "x_feats"), (2, "mask_preds"), (4, "seg_preds"), (5, "depth_feats"), (7, "depth_pred")):
        if precision in ("bf16", "fp16"):              # one plane: the first use of an fp32 map is the same rounding
            assert torch.equal(a[i], b[i]), name
        else:
            assert Hh.rel_err(a[i].cpu(), b[i].cpu()) < 2e-5, name
