"""GPU parity proper: the HIP path, called through the drop-in modules / the C ABI, against
(a) golden vectors produced by the reference itself and (b) the CPU oracle on seeded inputs.

Tolerances (BASELINE.json north_star: 1e-3 relative fp32, relative = max|diff| / max|ref|):
  * precision "fp32" (split-bf16 MFMA, fp32 accumulate): 1e-3 for every per-stage output with
    teacher-forced inputs -- the contract; observed ~1e-5.
  * precision "bf16" (the benchmark precision of BASELINE config 2): 3e-2 against the same oracle
    (bf16 has 8 mantissa bits; LayerNorm keeps errors relative).  Documented, not the 1e-3 gate.
  * free-running S stages: the hard threshold sigmoid(m) > 0.5 inside the recurrence can flip
    pixels on rounding differences; we report the flip rate and bound the final error
    loosely; the golden free-running fixture is additionally checked at 1e-3 in fp32 mode.
"""
import json

import numpy as np
import pytest
import torch

import helpers as Hh
from oracle import poly_oracle as O
from helpers import stage_cfg
from polyphonicformer_amd import _lib, engine as E
from polyphonicformer_amd.registry import HEADS, TRANSFORMER_LAYER
import polyphonicformer_amd.kernel_update  # noqa: F401  (registers the heads)
import polyphonicformer_amd.kernel_update_head  # noqa: F401
import polyphonicformer_amd.kernel_updator  # noqa: F401

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
TOL = {"fp32": 1e-3, "bf16": 3e-2}


def _full_weights():
    with open(Hh.GOLDEN + "/full_state_keys.json") as f:
        shapes = json.load(f)
    return Hh.seeded_fill(shapes, 1234)


def _iter_head(sd, S=3, n_thing=8, n_stuff=11, Nq=100, precision="fp32"):
    L = n_thing + n_stuff
    h = HEADS.build(dict(type="KernelUpdateIterHead", num_stages=S, assign_stages=S, stage_loss_weights=[1] * S,
                         num_proposals=Nq, num_thing_classes=n_thing, num_stuff_classes=n_stuff, do_panoptic=True,
                         merge_joint=True, mask_head=stage_cfg(256, 2048, 8, L, n_thing, n_stuff),
                         test_cfg=dict(max_per_img=Nq, mask_thr=0.5,
                                       merge_stuff_thing=dict(overlap_thr=0.6, instance_score_thr=0.3))))
    h.load_state_dict({k[len("roi_head."):]: v for k, v in sd.items() if k.startswith("roi_head.")
                       and int(k.split(".")[2]) < S})
    h.eval().to("cuda:0")
    h.set_precision(precision)
    return h


@pytest.fixture(scope="module")
def weights():
    return _full_weights()


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_kernel_updator_golden(gpu, weights, precision):
    z = Hh.load_golden("full_updator.npz")
    ku = TRANSFORMER_LAYER.build(dict(type="KernelUpdator", in_channels=256, feat_channels=256, out_channels=256,
                                      input_feat_shape=3, act_cfg=dict(type="ReLU", inplace=True),
                                      norm_cfg=dict(type="LN")))
    pre = "roi_head.mask_head.0.kernel_update_conv."
    ku.load_state_dict({k[len(pre):]: v for k, v in weights.items() if k.startswith(pre)})
    ku.to(gpu)
    ku.precision = precision
    u, k = torch.from_numpy(z["u"]).to(gpu), torch.from_numpy(z["k"]).to(gpu)
    out = ku(u, k[:, :, None, :])
    B, N = z["u"].shape[:2]
    assert out.shape == (B * N, 1, 256)
    e = Hh.rel_err(out.cpu().reshape(B, N, 256), z["out"])
    assert e < TOL[precision], e


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_stage_teacher_forced_golden(gpu, weights, precision):
    """KernelUpdateHead.forward per stage on the reference's own stage inputs."""
    z = Hh.load_golden("full_iter.npz")
    m = json.loads(bytes(z["meta_json"]).decode())
    cfg = m["cfg"]
    inp = Hh.iter_inputs(m["iseed"], m["B"], m["N"], 256, m["H"], m["W"])
    head = _iter_head(weights, cfg["S"], precision=precision)
    x, dfe = inp["x"].to(gpu), inp["dfe"].to(gpu)
    B, N = m["B"], m["N"]
    worst = {}
    for s in range(cfg["S"]):
        k = torch.from_numpy(z[f"s{s}_in_k"]).to(gpu).reshape(B, N, 256, 1, 1)
        q = torch.from_numpy(z[f"s{s}_in_q"]).to(gpu).reshape(B, N, 256, 1, 1)
        mp = torch.from_numpy(z[f"s{s}_in_m"]).to(gpu)
        cls, nm, obj, nd, dobj = head.mask_head[s](x, k, mp, depth_proposal=q, depth_feats=dfe)
        got = dict(cls=cls, mask=nm, obj=obj.reshape(B, N, 256), depth=nd, dobj=dobj.reshape(B, N, 256))
        for name, t in got.items():
            e = Hh.rel_err(t.cpu(), z[f"s{s}_{name}"])
            worst[(s, name)] = e
            assert e < TOL[precision], (s, name, e)
    print("teacher-forced rel err", precision, {k: f"{v:.1e}" for k, v in worst.items()})


def test_iter_free_running_golden(gpu, weights):
    """simple_test_mask_preds, S=3, fp32 mode, against the reference's free-running outputs."""
    z = Hh.load_golden("full_iter.npz")
    m = json.loads(bytes(z["meta_json"]).decode())
    S, B, N = m["cfg"]["S"], m["B"], m["N"]
    inp = {k: v.to(gpu) for k, v in Hh.iter_inputs(m["iseed"], B, N, 256, m["H"], m["W"]).items()}
    inp["q0"] = inp["q0"][:1, :1].expand(B, N, 256, 1, 1)      # stride-0 view on the device, like kernel_head.py:336
    head = _iter_head(weights, S, precision="fp32")
    metas = [Hh.img_meta(m["H"] * 8, m["W"] * 8)] * B
    assert not inp["q0"].is_contiguous()
    obj, cls, mask, mask_up = head.simple_test_mask_preds(inp["x"], inp["k0"], inp["m0"], None, metas,
                                                          depth_preds=inp["depth_pred"], depth_feats=inp["dfe"],
                                                          depth_proposal=inp["q0"])
    assert obj.shape == (B, N, 256, 1, 1) and mask_up.shape == (B, N, 2 * m["H"], 2 * m["W"])
    flips = ((mask.cpu() > 0) != (torch.from_numpy(z[f"s{S - 1}_mask"]) > 0)).float().mean().item()
    print("free-running sign flip rate of final mask logits:", flips)
    assert flips < 1e-3
    plan = next(iter(head._plans.values()))
    if flips == 0:                              # no hard decision differed: the reference's own outputs, at the 1e-3 contract
        assert Hh.rel_err(obj.cpu().reshape(B, N, 256), z["final_obj"]) < 1e-3
        assert Hh.rel_err(cls.cpu(), z["final_cls"]) < 1e-3
        assert Hh.rel_err(mask.cpu(), z[f"s{S - 1}_mask"]) < 1e-3
        assert Hh.rel_err(mask_up.cpu(), z["mask_up"]) < 1e-3
        assert Hh.rel_err(plan.depth_up.cpu(), z["depth_up"]) < 1e-3
    # whatever flipped: the same call again with the hard masks of the DEVICE run recorded, the oracle following them
    # (no blanket 5e-2) -- every output at 1e-3
    plan.debug_bits = []
    obj, cls, mask, mask_up = head.simple_test_mask_preds(inp["x"], inp["k0"], inp["m0"], None, metas,
                                                          depth_preds=inp["depth_pred"], depth_feats=inp["dfe"],
                                                          depth_proposal=inp["q0"])
    torch.cuda.synchronize()
    plan = next(iter(head._plans.values()))
    hard, plan.debug_bits = Hh.unpack_hard_masks(plan.debug_bits, N, m["H"], m["W"]), None
    assert len(hard) == S
    sd = {k[len("roi_head."):]: v for k, v in weights.items() if k.startswith("roi_head.")}
    ci = Hh.iter_inputs(m["iseed"], B, N, 256, m["H"], m["W"])
    refc = O.iter_head_mask_preds(sd, S, ci["x"], ci["k0"], ci["m0"], ci["q0"], ci["dfe"], hard_masks=hard)
    ec = {n: Hh.rel_err(t.cpu(), r) for n, t, r in (("obj", obj.reshape(B, N, 256), refc["obj"]), ("cls", cls, refc["cls"]),
                                                      ("mask", mask, refc["mask"]), ("mask_up", mask_up, refc["mask_up"]),
                                                      ("depth_up", plan.depth_up, refc["depth_up"]))}
    print("free running vs the oracle on the device's hard masks:", {k: f"{v:.1e}" for k, v in ec.items()})
    assert max(ec.values()) < 1e-3, ec
    # element-wise as well: |a - b| <= 1e-3 |b| + 0.75e-3 max|b| on EVERY element (fp32 mode: far inside)
    at = {n: Hh.needed_atol(t.cpu(), r, 1e-3) for n, t, r in (("obj", obj.reshape(B, N, 256), refc["obj"]), ("cls", cls, refc["cls"]),
                                                              ("mask", mask, refc["mask"]), ("mask_up", mask_up, refc["mask_up"]),
                                                              ("depth_up", plan.depth_up, refc["depth_up"]))}
    assert max(at.values()) < 0.75e-3, at


@pytest.mark.parametrize("precision,N,H,W,B", [("fp32", 153, 16, 24, 1), ("fp32", 40, 6, 13, 3), ("fp32", 253, 6, 26, 1),
                                                 ("bf16", 153, 16, 24, 2), ("bf16", 111, 9, 20, 1)])
def test_stage_vs_oracle_random_shapes(gpu, weights, precision, N, H, W, B):
    """one stage, ragged sizes (HW not a multiple of 128, N not a multiple of 32), oracle as checker."""
    head = _iter_head(weights, 1, precision=precision)
    inp = Hh.iter_inputs(1000 + N, B, N, 256, H, W, mask_bias=-0.5)
    sd = {k[len("roi_head."):]: v for k, v in weights.items()}
    ref = O.update_stage(sd, "mask_head.0.", inp["x"], inp["k0"].reshape(B, N, 256), inp["m0"],
                         inp["q0"].reshape(B, N, 256), inp["dfe"])
    g = {k: v.to(gpu) for k, v in inp.items()}
    cls, nm, obj, nd, dobj = head.mask_head[0](g["x"], g["k0"], g["m0"], depth_proposal=g["q0"], depth_feats=g["dfe"])
    for name, t, r in (("cls", cls, ref["cls"]), ("mask", nm, ref["mask"]), ("obj", obj.reshape(B, N, 256), ref["obj"]),
                       ("depth", nd, ref["depth"]), ("dobj", dobj.reshape(B, N, 256), ref["dobj"])):
        e = Hh.rel_err(t.cpu(), r)
        assert e < TOL[precision], (name, e)


def test_iter_bf16_vs_oracle_flip_rate(gpu, weights):
    """bf16 free-running S=3: report the binarisation flip rate, bound the outputs loosely."""
    B, N, H, W, S = 2, 111, 16, 32, 3
    inp = Hh.iter_inputs(4242, B, N, 256, H, W)
    sd = {k[len("roi_head."):]: v for k, v in weights.items()}
    ref = O.iter_head_mask_preds(sd, S, inp["x"], inp["k0"], inp["m0"], inp["q0"], inp["dfe"])
    head = _iter_head(weights, S, precision="bf16")
    g = {k: v.to(gpu) for k, v in inp.items()}
    obj, cls, mask, mask_up = head.simple_test_mask_preds(g["x"], g["k0"], g["m0"], None, [Hh.img_meta(H * 8, W * 8)] * B,
                                                          depth_preds=g["depth_pred"], depth_feats=g["dfe"],
                                                          depth_proposal=g["q0"])
    flips = ((mask.cpu() > 0) != (ref["mask"] > 0)).float().mean().item()
    print("bf16 free-running flip rate:", flips, "obj rel err:", Hh.rel_err(obj.cpu().reshape(B, N, 256), ref["obj"]))
    assert flips < 0.05
    assert Hh.rel_err(obj.cpu().reshape(B, N, 256), ref["obj"]) < 0.1
    # determinism: a second run is bit identical (fixed-order split-K reduction, no atomics)
    obj2, cls2, mask2, mask_up2 = head.simple_test_mask_preds(g["x"], g["k0"], g["m0"], None, [Hh.img_meta(H * 8, W * 8)] * B,
                                                              depth_preds=g["depth_pred"], depth_feats=g["dfe"],
                                                              depth_proposal=g["q0"])
    assert torch.equal(mask_up, mask_up2) and torch.equal(obj, obj2)


# ---------------------------------------------------------------------------------------------------
#  KernelHead (a1) and the whole a1 -> a6 path
# ---------------------------------------------------------------------------------------------------
import polyphonicformer_amd.kernel_head  # noqa: E402,F401


def _kernel_head(sd, precision="fp32", n_thing=8, n_stuff=11, Nq=100):
    h = HEADS.build(dict(type="KernelHead", num_proposals=Nq, num_classes=n_thing + n_stuff, num_thing_classes=n_thing,
                         num_stuff_classes=n_stuff, in_channels=256, out_channels=256, cat_stuff_mask=True,
                         feat_downsample_stride=2, feat_refine_stride=1, feat_refine=False, use_binary=True,
                         conv_normal_init=True, proposal_feats_with_obj=True, xavier_init_kernel=False, kernel_init_std=1,
                         loss_seg=dict(type="FocalLoss", use_sigmoid=True), loss_mask=dict(type="CrossEntropyLoss", use_sigmoid=True),
                         localization_fpn=None))
    h.load_state_dict({k[len("rpn_head."):]: v for k, v in sd.items() if k.startswith("rpn_head.")})
    h.eval().to("cuda:0")
    h.set_precision(precision)
    return h


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_kernel_head_golden(gpu, weights, precision):
    z = Hh.load_golden("full_khead.npz")
    m = json.loads(bytes(z["meta_json"]).decode())
    B, N = m["B"], m["N"]
    feats = [f.to(gpu) for f in Hh.neck_inputs(m["nseed"], B, 256, m["H"], m["W"])]
    kh = _kernel_head(weights, precision)
    (pf, xf, mp, cs, seg, df, dp, dpr, aspp) = kh.simple_test_rpn(feats, [Hh.img_meta(m["H"] * 8, m["W"] * 8)] * B)
    assert cs is None and aspp is None and not dp.is_contiguous() and dp.shape == (B, N, 256, 1, 1)
    tol = TOL[precision]
    for name, t in (("x_feats", xf), ("mask_preds", mp), ("seg_preds", seg), ("depth_feats", df), ("depth_pred", dpr)):
        e = Hh.rel_err(t.cpu(), z[name])
        assert e < tol, (name, e)
    assert Hh.rel_err(pf.cpu().reshape(B, N, 256), z["proposal_feats"]) < tol
    assert Hh.rel_err(dp.cpu().reshape(B, N, 256), z["depth_proposal"]) < 1e-7


@pytest.mark.parametrize("precision", ["fp32", "fp16"])
@pytest.mark.parametrize("H,W,B", [(6, 13, 2), (16, 24, 1)])
def test_kernel_head_vs_oracle_ragged(gpu, weights, H, W, B, precision):
    feats = Hh.neck_inputs(321, B, 256, H, W)
    sd = {k[len("rpn_head."):]: v for k, v in weights.items() if k.startswith("rpn_head.")}
    ref = O.kernel_head_post_neck(sd, *feats, 8, 19, 32)
    kh = _kernel_head(weights, precision)
    out = kh.simple_test_rpn([f.to(gpu) for f in feats], [Hh.img_meta(H * 8, W * 8)] * B)
    N = 111
    for name, t in (("x_feats", out[1]), ("mask_preds", out[2]), ("seg_preds", out[4]), ("depth_feats", out[5]),
                    ("depth_pred", out[7])):
        assert Hh.rel_err(t.cpu(), ref[name]) < 1e-3, name
    # proposal_feats = init_kernels + pool(binarise(mask logits), x): a logit within rounding of 0 may binarise
    # differently than in the oracle and moves a whole feature vector (SURVEY.md 7 "hard threshold").  Check the
    # pooling against the masks the device actually produced, and bound the number of flipped pixels.
    mp, xf = out[2].cpu(), out[1].cpu()
    flips = int(((mp[:, :100] > 0) != (ref["mask_preds"][:, :100] > 0)).sum())
    print("KernelHead binarisation flips vs oracle:", flips, "of", mp[:, :100].numel())
    assert flips <= (4 if precision == "fp32" else 16)
    own = sd["init_kernels.weight"].reshape(1, 100, 256) + torch.einsum("bnhw,bchw->bnc", (mp[:, :100] > 0).float(), xf)
    assert Hh.rel_err(out[0].cpu().reshape(B, N, 256)[:, :100], own) < 1e-3
    assert Hh.rel_err(out[0].cpu().reshape(B, N, 256)[:, 100:], ref["proposal_feats"].reshape(B, N, 256)[:, 100:]) < 1e-6
    if flips == 0:
        assert Hh.rel_err(out[0].cpu().reshape(B, N, 256), ref["proposal_feats"].reshape(B, N, 256)) < 1e-3


@pytest.mark.parametrize("precision,Nq", [("bf16", 200), ("fp32", 200), ("bf16", 37)])
def test_kernel_head_row_counts(gpu, precision, Nq):
    """init_kernels with 7 row tiles (cfg5's 200 proposals: the fused second GEMM reads its fragments from L2 instead of
    LDS) and with a ragged 37 rows, random-init weights, against the oracle."""
    torch.manual_seed(5)
    n_thing, n_stuff = 8, 11
    h = HEADS.build(dict(type="KernelHead", num_proposals=Nq, num_classes=n_thing + n_stuff, num_thing_classes=n_thing,
                         num_stuff_classes=n_stuff, in_channels=256, out_channels=256, cat_stuff_mask=True,
                         feat_downsample_stride=2, feat_refine_stride=1, feat_refine=False, use_binary=True,
                         conv_normal_init=True, proposal_feats_with_obj=True, xavier_init_kernel=False, kernel_init_std=1,
                         loss_seg=dict(type="FocalLoss", use_sigmoid=True), loss_mask=dict(type="CrossEntropyLoss", use_sigmoid=True),
                         localization_fpn=None))
    h.init_weights()
    sd = {k: v.detach().clone() for k, v in h.state_dict().items()}
    h.eval().to(gpu)
    h.set_precision(precision)
    B, H, W = 2, 12, 20
    feats = Hh.neck_inputs(77, B, 256, H, W)
    ref = O.kernel_head_post_neck(sd, *feats, n_thing, n_thing + n_stuff, 32)
    out = h.simple_test_rpn([f.to(gpu) for f in feats], [Hh.img_meta(H * 8, W * 8)] * B)
    tol = TOL[precision]
    for name, t in (("x_feats", out[1]), ("mask_preds", out[2]), ("seg_preds", out[4]), ("depth_feats", out[5]),
                    ("depth_pred", out[7])):
        assert t.shape == ref[name].shape, name
        assert Hh.rel_err(t.cpu(), ref[name]) < tol, (name, Hh.rel_err(t.cpu(), ref[name]))


def test_kernel_head_without_stuff_rows(gpu, weights):
    """cat_stuff_mask=False (kernel_head.py:329 skipped): N = num_proposals, no dual store of the stuff logits"""
    sd = {k[len("rpn_head."):]: v for k, v in weights.items() if k.startswith("rpn_head.")}
    h = HEADS.build(dict(type="KernelHead", num_proposals=100, num_classes=19, num_thing_classes=8, num_stuff_classes=11,
                         in_channels=256, out_channels=256, cat_stuff_mask=False, feat_downsample_stride=2, feat_refine=False,
                         use_binary=True, conv_normal_init=True, proposal_feats_with_obj=True, kernel_init_std=1,
                         loss_seg=dict(type="FocalLoss", use_sigmoid=True), localization_fpn=None))
    h.load_state_dict(sd)
    h.eval().to(gpu)
    h.set_precision("fp32")
    B, H, W = 2, 8, 24
    feats = Hh.neck_inputs(91, B, 256, H, W)
    ref = O.kernel_head_post_neck(sd, *feats, 8, 19, 32, cat_stuff_mask=False)
    out = h.simple_test_rpn([f.to(gpu) for f in feats], [Hh.img_meta(H * 8, W * 8)] * B)
    assert out[2].shape == (B, 100, H, W) and out[0].shape == (B, 100, 256, 1, 1) and out[6].shape[1] == 1
    for name, t in (("x_feats", out[1]), ("mask_preds", out[2]), ("seg_preds", out[4]), ("depth_feats", out[5]),
                    ("depth_pred", out[7])):
        assert Hh.rel_err(t.cpu(), ref[name]) < 1e-3, name
    flips = int(((out[2].cpu() > 0) != (ref["mask_preds"] > 0)).sum())
    if flips == 0:
        assert Hh.rel_err(out[0].cpu().reshape(B, 100, 256), ref["proposal_feats"].reshape(B, 100, 256)) < 1e-3


def test_whole_path_a1_a6(gpu, weights):
    """KernelHead -> KernelUpdateIterHead exactly as Polyphonic.simple_test wires them
    (polyphonic_former.py:145-161), with the plane/bit hand-off, vs the oracle's run_head."""
    B, H, W, S = 2, 8, 16, 3
    feats = Hh.neck_inputs(99, B, 256, H, W)
    ref = O.run_head(weights, feats, S, 8, 19)
    kh, ih = _kernel_head(weights, "fp32"), _iter_head(weights, S, precision="fp32")
    metas = [Hh.img_meta(H * 8, W * 8)] * B
    (pf, xf, mp, cs, seg, df, dp, dpr, _) = kh.simple_test_rpn([f.to(gpu) for f in feats], metas)
    assert hasattr(xf, "_ph_handoff")
    obj, cls, mask, mask_up = ih.simple_test_mask_preds(xf, pf, mp, cs, metas, depth_preds=dpr, depth_feats=df, depth_proposal=dp)
    plan = next(iter(ih._plans.values()))
    assert plan.handoff_runs == 1                   # the hand-off path ran (no ingest)
    flips = ((mask.cpu() > 0) != (ref["mask"] > 0)).float().mean().item()
    print("a1->a6 free-running flip rate:", flips)
    # free running through 1 + 3 hard thresholds: a logit within rounding of 0 binarises differently and moves a whole
    # feature vector; the flip rate is bounded, and the arithmetic is compared at 1e-3 with the oracle
    # following the hard masks of the DEVICE run (a1's pooling uses the thing rows of the first one) -- no blanket 5e-2
    assert flips < 1e-3
    plan.debug_bits = []
    (pf, xf, mp, cs, seg, df, dp, dpr, _) = kh.simple_test_rpn([f.to(gpu) for f in feats], metas)
    obj, cls, mask, mask_up = ih.simple_test_mask_preds(xf, pf, mp, cs, metas, depth_preds=dpr, depth_feats=df, depth_proposal=dp)
    torch.cuda.synchronize()
    plan = next(iter(ih._plans.values()))
    hard, plan.debug_bits = Hh.unpack_hard_masks(plan.debug_bits, mask.shape[1], H, W), None
    assert len(hard) == S
    refc = O.run_head(weights, feats, S, 8, 19, hard_masks=hard)
    ec = {n: Hh.rel_err(t.cpu(), r) for n, t, r in (("obj", obj.reshape(B, -1, 256), refc["obj"]), ("cls", cls, refc["cls"]),
                                                      ("mask", mask, refc["mask"]), ("mask_up", mask_up, refc["mask_up"]),
                                                      ("depth_up", plan.depth_up, refc["depth_up"]))}
    print("a1->a6 vs the oracle on the device's hard masks:", {k: f"{v:.1e}" for k, v in ec.items()})
    assert max(ec.values()) < 1e-3, ec
    if flips == 0:
        assert Hh.rel_err(obj.cpu().reshape(B, -1, 256), ref["obj"]) < 1e-3
        assert Hh.rel_err(cls.cpu(), ref["cls"]) < 1e-3
        assert Hh.rel_err(mask_up.cpu(), ref["mask_up"]) < 1e-3


def test_kernel_head_fp16_grade_and_handoff(gpu, weights):
    """KernelHead's third grade (`set_precision("fp16")`: ONE fp16 plane of the maps and weights, one MFMA per product):
    every output of the post-neck part within 1e-3 of the REFERENCE golden, and the fp16 planes + mask bits it hands to a
    KernelUpdateIterHead in `fp16` mode decode to exactly what the fp32 tensors of the same call decode to."""
    z = Hh.load_golden("full_khead.npz")
    m = json.loads(bytes(z["meta_json"]).decode())
    B, N, H, W = m["B"], m["N"], m["H"], m["W"]
    feats = [f.to(gpu) for f in Hh.neck_inputs(m["nseed"], B, 256, H, W)]
    kh = _kernel_head(weights, "fp16")
    metas = [Hh.img_meta(H * 8, W * 8)] * B
    (pf, xf, mp, cs, seg, df, dp, dpr, _) = kh.simple_test_rpn(feats, metas)
    for name, t in (("x_feats", xf), ("mask_preds", mp), ("seg_preds", seg), ("depth_feats", df), ("depth_pred", dpr)):
        e = Hh.rel_err(t.cpu(), z[name])
        print("KernelHead fp16 grade", name, e)
        assert e < 1e-3, (name, e)
    flips = int(((mp.cpu() > 0) != (torch.from_numpy(z["mask_preds"]) > 0)).sum())
    assert flips <= 8, flips
    if flips == 0:
        assert Hh.rel_err(pf.cpu().reshape(B, N, 256), z["proposal_feats"]) < 1e-3
    ih = _iter_head(weights, 3, precision="fp16")
    a = ih.simple_test_mask_preds(xf, pf, mp, cs, metas, depth_preds=dpr, depth_feats=df, depth_proposal=dp)
    plan = next(iter(ih._plans.values()))
    assert plan.handoff_runs == 1                   # fp16 planes and bits adopted, no ingest / binarize
    a = [t.clone() for t in a]
    b = ih.simple_test_mask_preds(xf.clone(), pf, mp, cs, metas, depth_preds=dpr, depth_feats=df, depth_proposal=dp)   # no hand-off
    assert plan.handoff_runs == 1
    for t, u in zip(a, b):
        assert torch.equal(t, u)


@pytest.mark.parametrize("mode", ["mixed16", "mixed"])
def test_parity_grade_kernel_head_hands_hi_planes_to_the_mixed_modes(gpu, weights, mode):
    """KernelHead at the parity grade (hi + lo bf16 planes) -> KernelUpdateIterHead in a mode that reads ONE bf16 plane: the hi
    plane and the mask bits are adopted (no ingest / binarize pass) and decode to exactly what the fp32 tensors of the same
    call decode to through the ingest kernel"""
    B, H, W = 2, 8, 16
    feats = [f.to(gpu) for f in Hh.neck_inputs(41, B, 256, H, W)]
    kh = _kernel_head(weights, "fp32")
    metas = [Hh.img_meta(H * 8, W * 8)] * B
    (pf, xf, mp, cs, seg, df, dp, dpr, _) = kh.simple_test_rpn(feats, metas)
    ih = _iter_head(weights, 3, precision=mode)
    ih.set_precision(mode, torch.float16)
    a = [t.clone() for t in ih.simple_test_mask_preds(xf, pf, mp, cs, metas, depth_preds=dpr, depth_feats=df, depth_proposal=dp)]
    plan = next(iter(ih._plans.values()))
    assert plan.handoff_runs == 1
    b = ih.simple_test_mask_preds(xf.clone(), pf, mp, cs, metas, depth_preds=dpr, depth_feats=df, depth_proposal=dp)   # no hand-off
    assert plan.handoff_runs == 1
    for t, u in zip(a, b):
        assert torch.equal(t, u)


def test_bf16_feature_inputs_skip_ingest(gpu, weights):
    """bf16 NCHW feature tensors are the plane format: same result as fp32 inputs rounded by the ingest kernel"""
    B, N, H, W, S = 1, 111, 8, 16, 2
    inp = {k: v.to(gpu) for k, v in Hh.iter_inputs(77, B, N, 256, H, W).items()}
    head = _iter_head(weights, S, precision="bf16")
    metas = [Hh.img_meta(H * 8, W * 8)] * B
    a = head.simple_test_mask_preds(inp["x"], inp["k0"], inp["m0"], None, metas, depth_feats=inp["dfe"], depth_proposal=inp["q0"])
    a = [t.clone() for t in a]
    b = head.simple_test_mask_preds(inp["x"].to(torch.bfloat16), inp["k0"], inp["m0"], None, metas,
                                    depth_feats=inp["dfe"].to(torch.bfloat16), depth_proposal=inp["q0"])
    for t, u in zip(a, b):
        assert torch.equal(t, u)
    head.set_precision("fp32")
    with pytest.raises(_lib.PolyheadError):
        head.simple_test_mask_preds(inp["x"].to(torch.bfloat16), inp["k0"], inp["m0"], None, metas,
                                    depth_feats=inp["dfe"].to(torch.bfloat16), depth_proposal=inp["q0"])


def test_cfg1_exact_shape(gpu):
    """BASELINE configs[0] at its exact shape (one 256x512 frame -> 32x64, N = 100 + 11, S = 1) on the HIP path, KernelHead ->
    simple_test_mask_preds as Polyphonic.simple_test wires them, fp32 grade: against the REFERENCE's outputs (tests/golden/cfg1.npz)
    where no hard decision differs, and against the oracle following the device's hard masks at 1e-3 either way."""
    from test_oracle_golden import cfg1_case
    m, sd, feats, z = cfg1_case()
    cfg = m["cfg"]
    B, N, H, W = m["B"], m["N"], m["H"], m["W"]
    kh, ih = _kernel_head(sd, "fp32"), _iter_head(sd, 1, precision="fp32")
    metas = [Hh.img_meta(H * 8, W * 8)]
    plan = ih._plan(B, N, H, W, torch.device("cuda:0"))
    plan.debug_bits = []
    (pf, xf, mp, cs, seg, df, dp, dpr, _) = kh.simple_test_rpn([f.to(gpu) for f in feats], metas)
    obj, cls, mask, mask_up = ih.simple_test_mask_preds(xf, pf, mp, cs, metas, depth_preds=dpr, depth_feats=df, depth_proposal=dp)
    torch.cuda.synchronize()
    plan = next(iter(ih._plans.values()))
    hard, plan.debug_bits = Hh.unpack_hard_masks(plan.debug_bits, N, H, W), None
    assert Hh.rel_err(mp.cpu(), z["kh_mask_preds"]) < 1e-3 and Hh.rel_err(dpr.cpu(), z["kh_depth_pred"]) < 1e-3
    flips = float((hard[0] != O.binarize(torch.from_numpy(z["kh_mask_preds"]))).float().mean())
    print("cfg1: flip rate of the first hard mask vs the reference's:", flips)
    assert flips < 1e-3
    refc = O.run_head(sd, feats, 1, cfg["n_thing"], cfg["n_thing"] + cfg["n_stuff"], cfg["heads"], cfg["groups"], hard_masks=hard)
    ec = {n: Hh.rel_err(t.cpu(), r) for n, t, r in (("obj", obj.reshape(B, N, 256), refc["obj"]), ("cls", cls, refc["cls"]),
                                                      ("mask", mask, refc["mask"]), ("mask_up", mask_up, refc["mask_up"]),
                                                      ("depth_up", plan.depth_up, refc["depth_up"]))}
    print("cfg1 vs the oracle on the device's hard masks:", {k: f"{v:.1e}" for k, v in ec.items()})
    assert max(ec.values()) < 1e-3, ec
    if flips == 0:
        eg = dict(obj=Hh.rel_err(obj.cpu().reshape(B, N, 256), z["obj"]), cls=Hh.rel_err(cls.cpu(), z["cls"]),
                  mask=Hh.rel_err(mask.cpu(), z["mask"]), mask_up=Hh.rel_err(mask_up.cpu()[..., 0::3, 0::3], z["mask_up_s"]),
                  depth_up=Hh.rel_err(plan.depth_up.cpu()[..., 0::3, 0::3], z["depth_up_s"]))
        print("cfg1 vs the reference's golden:", {k: f"{v:.1e}" for k, v in eg.items()})
        assert max(eg.values()) < 1e-3, eg
