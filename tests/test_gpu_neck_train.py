"""GPU: SemanticFPNWrapper in TRAINING (VERDICT r04 #1c) -- the differentiable form of the neck (`train.neck_forward_train`:
3x3 conv / GroupNorm + ReLU / x2 upsample nodes of libpolyhead with hand-written backward) against the REFERENCE's own class under
torch autograd (tests/golden/neck_train*.npz, oracle/gen_golden_neck.py): the three outputs, the gradient of all 30 parameter
tensors and of the four FPN inputs.  Plus the 3x3 convolution node alone against torch's conv2d autograd on the CPU."""
import json

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import helpers as Hh
from polyphonicformer_amd import train as T
from polyphonicformer_amd.registry import NECKS
import polyphonicformer_amd.semantic_fpn  # noqa: F401

pytestmark = pytest.mark.gpu


def _digest(t, n):
    f = t.detach().double().reshape(-1).cpu()
    idx = torch.linspace(0, f.numel() - 1, n).long()
    return np.concatenate([[float(f.norm()), float(f.sum())], f[idx].numpy()])


def _cmp(got, ref):
    e_norm = abs(got[0] - ref[0]) / max(ref[0], 1e-30)
    e_ent = float(np.abs(got[2:] - ref[2:]).max() / max(np.abs(ref[2:]).max(), 1e-30))
    return e_norm, e_ent


@pytest.mark.parametrize("stride,B,K,M,H,W", [(1, 2, 256, 256, 9, 14), (2, 1, 256, 256, 10, 13), (1, 1, 32, 48, 5, 7), (2, 2, 32, 48, 7, 7)])
def test_conv3x3_node_vs_torch(gpu, stride, B, K, M, H, W):
    g = torch.Generator().manual_seed(H * W + stride)
    x, w = torch.randn(B, K, H, W, generator=g), torch.randn(M, K, 3, 3, generator=g) * 0.05
    with torch.enable_grad():
        xc, wc = x.double().requires_grad_(True), w.double().requires_grad_(True)
        yc = F.conv2d(xc, wc, None, stride=stride, padding=1)
        cot = torch.randn(yc.shape, generator=g)
        (yc * cot.double()).sum().backward()
        xd, wd = x.to(gpu).requires_grad_(True), w.to(gpu).requires_grad_(True)
        yd = T._Conv3x3.apply(xd, wd, stride)
        (yd * cot.to(gpu)).sum().backward()
    torch.cuda.synchronize()
    assert yd.shape == yc.shape
    e = (Hh.rel_err(yd.detach().cpu(), yc.detach()), Hh.rel_err(xd.grad.cpu(), xc.grad), Hh.rel_err(wd.grad.cpu(), wc.grad))
    print("conv3x3 node (y, dx, dw) rel err:", e)
    assert max(e) < 3e-5, e


@pytest.mark.parametrize("golden", ["neck_train.npz", "neck_train_b.npz"])
def test_neck_training_vs_reference(gpu, golden):
    z = Hh.load_golden(golden)
    m = json.loads(bytes(z["meta_json"]).decode())
    neck = NECKS.build(dict(type="SemanticFPNWrapper", in_channels=256, feat_channels=256, out_channels=256, start_level=0, end_level=3,
                            upsample_times=2, positional_encoding=dict(type="SinePositionalEncoding", num_feats=m["nf"], normalize=True),
                            cat_coors=False, cat_coors_level=3, fuse_by_cat=False, return_list=False, num_aux_convs=2,
                            norm_cfg=dict(type="GN", num_groups=m["groups"], requires_grad=True)))
    with open(Hh.GOLDEN + "/neck_state_keys.json") as f:
        shapes = json.load(f)["full"]
    neck.load_state_dict(Hh.seeded_fill({k: tuple(v) for k, v in shapes.items()}, m["wseed"]))
    neck.to(gpu).train()
    feats = [f.to(gpu).requires_grad_(True) for f in Hh.fpn_inputs(seed=m["iseed"], B=m["B"], C=m["C"], H0=m["H0"], W0=m["W0"])]
    g = torch.Generator().manual_seed(m["cseed"])
    with torch.enable_grad():
        outs = neck(feats)
        cots = [torch.randn(o.shape, generator=g) for o in outs]
        sum((o * c.to(gpu)).sum() for o, c in zip(outs, cots)).backward()
    torch.cuda.synchronize()
    for j, o in enumerate(outs):
        e = _cmp(_digest(o, 4096), z[f"out{j}"])
        print("output", j, e)
        assert e[0] < 1e-4 and e[1] < 1e-4, (j, e)
    worst = ("", 0.0)
    for n, p in neck.named_parameters():
        assert p.grad is not None, n
        e = _cmp(_digest(p.grad, 256), z["g_" + n])
        if max(e) > worst[1]:
            worst = (n, max(e))
        assert e[0] < 1e-3 and e[1] < 1e-3, (n, e)
    for i, f in enumerate(feats):
        e = _cmp(_digest(f.grad, 4096), z[f"gin{i}"])
        print("d / d FPN level", i, e)
        assert e[0] < 1e-3 and e[1] < 1e-3, (i, e)
    print("neck parameter gradients vs the reference's autograd: worst", worst)
    # inference form (no autograd) still runs the packed 16-bit kernels and agrees with the training form's outputs at its own grade
    with torch.no_grad():
        inf = neck([f.detach() for f in feats])
    for a, b in zip(inf, outs):
        assert Hh.rel_err(a.float().cpu(), b.detach().cpu()) < 1e-3


def test_kernel_head_trains_the_neck(gpu):
    """KernelHead.forward_train with this package's neck as localization_fpn (what polyphonic_former.py:97-110 runs): the loss
    dict's backward leaves gradients on every neck parameter and on the four FPN inputs -- the `frozen_neck_ok` refusal is gone"""
    from polyphonicformer_amd.registry import HEADS
    import polyphonicformer_amd.kernel_head  # noqa: F401
    neckc = dict(type="SemanticFPNWrapper", in_channels=256, feat_channels=256, out_channels=256, start_level=0, end_level=3,
                 upsample_times=2, positional_encoding=dict(type="SinePositionalEncoding", num_feats=128, normalize=True),
                 cat_coors=False, cat_coors_level=3, fuse_by_cat=False, return_list=False, num_aux_convs=2,
                 norm_cfg=dict(type="GN", num_groups=32, requires_grad=True))
    tc = dict(assigner=dict(type="MaskHungarianAssignerWithDepth", cls_cost=dict(type="FocalLossCost", weight=2.0),
                            dice_cost=dict(type="DiceCost", weight=4.0, pred_act=True), mask_cost=dict(type="MaskCost", weight=1.0, pred_act=True)),
              sampler=dict(type="MaskPseudoSampler"), pos_weight=1)
    kh = HEADS.build(dict(type="KernelHead", num_proposals=100, num_classes=19, num_thing_classes=8, num_stuff_classes=11,
                          cat_stuff_mask=True, feat_downsample_stride=2, feat_refine=False, use_binary=True, proposal_feats_with_obj=True,
                          kernel_init_std=1, conv_normal_init=True, loss_rank=dict(type="CrossEntropyLoss", use_sigmoid=False, loss_weight=0.1),
                          loss_seg=dict(type="FocalLoss", use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0),
                          loss_mask=dict(type="CrossEntropyLoss", use_sigmoid=True, loss_weight=1.0), loss_dice=dict(type="DiceLoss", loss_weight=4.0),
                          loss_depth=dict(type="DepthLoss", loss_weight=5.0, depth_act_mode="sigmoid"), localization_fpn=neckc, train_cfg=tc))
    kh.init_weights()
    kh.to(gpu).train()
    B, H0, W0 = 2, 16, 32
    fpn = [f.to(gpu).requires_grad_(True) for f in Hh.fpn_inputs(seed=3, B=B, C=256, H0=H0, W0=W0)]
    gts = [{k: v.to(gpu) for k, v in g_.items()} for g_ in Hh.train_gt(9, B, H0, W0, 8, 11, [3, 5])]
    metas = [Hh.img_meta(H0 * 4, W0 * 4)] * B
    gd = torch.stack([g_["depth"][None] for g_ in gts])
    with torch.enable_grad():
        out = kh.forward_train(fpn, metas, [g_["masks"] for g_ in gts], [g_["labels"] for g_ in gts], [g_["sem_seg"] for g_ in gts],
                               [g_["sem_cls"] for g_ in gts], gd)
        losses = out[0]
        sum(v.mean() for k_, v in losses.items() if "loss" in k_).backward()
    torch.cuda.synchronize()
    for n, p in kh.localization_fpn.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all() and float(p.grad.abs().max()) > 0, n
    for f in fpn:
        assert f.grad is not None and torch.isfinite(f.grad).all() and float(f.grad.abs().max()) > 0
