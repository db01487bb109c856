"""GPU: the device-backed mask Hungarian assigner (polyphonicformer_amd/assigner.py + csrc/ph_match.hip) against the
reference's goldens and the oracle: cost matrices within 1e-4 relative (fp32 contract 1e-3), integer assignments
identical."""
import numpy as np
import pytest
import torch

from tests import helpers as Hh
from oracle import assign_oracle as AO

pytestmark = pytest.mark.gpu

CFG = dict(type='MaskHungarianAssignerWithDepth', cls_cost=dict(type='FocalLossCost', weight=2.0),
           dice_cost=dict(type='DiceCost', weight=4.0, pred_act=True), mask_cost=dict(type='MaskCost', weight=1.0, pred_act=True),
           depth_cost=dict(type='DepthCost', weight=0., loss_fn=dict(type='DepthMatchLoss', loss_weight=1.),
                           depth_act_mode='sigmoid'))


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _dev(c, gpu):
    return {k: (v.to(gpu) if torch.is_tensor(v) else v) for k, v in c.items()}


@pytest.mark.parametrize("i", range(len(Hh.ASSIGN_CASES)))
def test_assigner_vs_reference_golden(gpu, i):
    from polyphonicformer_amd import assigner as A
    gold = Hh.load_golden("assign.npz")
    c = _dev(Hh.assign_case(**Hh.ASSIGN_CASES[i]), gpu)
    a = A.build_assigner(dict(CFG))
    sums = A.MatchSums(c["mask_logits"][None], c["gt_masks"][None], None if c["gt_valid"] is None else c["gt_valid"][None])
    cost = a.costs(sums, 0, c["cls_logits"], c["gt_labels"])
    e = Hh.rel_err(cost.cpu(), gold[f"c{i}_cost"])
    assert e < 1e-4, e
    r = a.assign(c["mask_logits"], c["cls_logits"], c["gt_masks"], c["gt_labels"], None, gt_valid=c["gt_valid"])
    assert r.num_gts == c["gt_masks"].shape[0] and r.gt_inds.is_cuda
    assert np.array_equal(r.gt_inds.cpu().numpy(), gold[f"c{i}_gt_inds"])
    assert np.array_equal(r.labels.cpu().numpy(), gold[f"c{i}_labels"])


@pytest.mark.parametrize("i", Hh.DEPTH_COST_CASES)
@pytest.mark.parametrize("mode", ["sigmoid", "monodepth"])
def test_depth_cost_vs_reference_golden(gpu, i, mode):
    """DepthCost with a non-zero weight (funcs/assigner.py:17-80) on ph_depth_cost_sums: the cost matrix against the
    reference's and the assignment it leads to (it differs from the depth-free one on three of the four fixtures)"""
    from polyphonicformer_amd import assigner as A
    gold = Hh.load_golden("assign.npz")
    case = Hh.ASSIGN_CASES[i]
    c = _dev(Hh.assign_case(**case), gpu)
    z, gd = (t.to(gpu) for t in Hh.assign_depth_inputs(case["seed"], case["N"], case["H"], case["W"]))
    cfg = dict(CFG, depth_cost=dict(type='DepthCost', weight=0.5, loss_fn=dict(type='DepthMatchLoss', loss_weight=1.), depth_act_mode=mode))
    a = A.build_assigner(cfg)
    dc = a.depth_cost(inputs=z, depth_gt=gd, target_masks=c["gt_masks"])
    e = Hh.rel_err(dc.cpu(), gold[f"d{i}_{mode}_depth_cost"])
    assert e < 1e-4, e
    r = a.assign(c["mask_logits"], c["cls_logits"], c["gt_masks"], c["gt_labels"], None, depth_pred=z, gt_depth=gd, gt_valid=c["gt_valid"])
    assert np.array_equal(r.gt_inds.cpu().numpy(), gold[f"d{i}_{mode}_gt_inds"])
    assert np.array_equal(r.labels.cpu().numpy(), gold[f"d{i}_{mode}_labels"])


def test_pixel_sums_vs_einsum(gpu):
    """every sum of the record against fp64 torch, ragged sizes (HW % 4 != 0), batch of 3 with zero-padded gt rows"""
    from polyphonicformer_amd import assigner as A
    g = torch.Generator().manual_seed(3)
    B, N, G, H, W = 3, 45, 9, 9, 13
    z = torch.randn(B, N, H, W, generator=g) * 3
    t = torch.rand(B, G, H, W, generator=g).round_(decimals=2)
    t[1, 5:] = 0
    v = (torch.rand(B, H, W, generator=g) > 0.3).float()
    s = A.MatchSums(z.to(gpu), t.to(gpu), v.to(gpu))
    p, td, vd = z.double().sigmoid(), t.double(), v.double()
    ref = dict(A=torch.einsum("bnhw,bghw,bhw->bng", p, td, vd), S=torch.einsum("bnhw,bhw->bn", p, vd),
               Q=torch.einsum("bnhw,bhw->bn", p * p, vd), C=torch.einsum("bghw,bhw->bg", td * td, vd),
               T=torch.einsum("bghw,bhw->bg", td, vd), V=vd.sum((1, 2)))
    for k, r in ref.items():
        assert Hh.rel_err(getattr(s, k).cpu().double(), r) < 2e-5, k
    s2 = A.MatchSums(z.to(gpu), t.to(gpu), None)                   # no gt_valid: V = H * W exactly
    assert torch.equal(s2.V.cpu(), torch.full((B,), float(H * W)))
    assert Hh.rel_err(s2.A.cpu().double(), torch.einsum("bnhw,bghw->bng", p, td)) < 2e-5


def test_assigner_full_size_vs_oracle(gpu):
    """training-crop size of the shipped config (512 x 1024 crop, assign stride 4 -> 128 x 256), N = 100, 40 instances"""
    from polyphonicformer_amd import assigner as A
    c = Hh.assign_case(seed=21, N=100, G=40, L=8, H=128, W=256)
    ref_inds, ref_labels = AO.assign(c["mask_logits"], c["cls_logits"], c["gt_masks"], c["gt_labels"], c["gt_valid"])
    d = _dev(c, gpu)
    r = A.build_assigner(dict(CFG)).assign(d["mask_logits"], d["cls_logits"], d["gt_masks"], d["gt_labels"], None,
                                           gt_valid=d["gt_valid"])
    assert torch.equal(r.gt_inds.cpu(), ref_inds) and torch.equal(r.labels.cpu(), ref_labels)


def test_assigner_without_depth_and_pids(gpu):
    from polyphonicformer_amd import assigner as A
    c = _dev(Hh.assign_case(seed=22, N=30, G=5, L=8, H=10, W=12, with_valid=False), gpu)
    a = A.build_assigner(dict(type='MaskHungarianAssigner', cls_cost=dict(type='FocalLossCost', weight=2.0),
                              dice_cost=dict(type='DiceCost', weight=4.0, pred_act=True),
                              mask_cost=dict(type='MaskCost', weight=1.0, pred_act=True)))
    pids = torch.arange(100, 105, device=gpu)
    r = a.assign(c["mask_logits"], c["cls_logits"], c["gt_masks"], c["gt_labels"], gt_pids=pids)
    ref_inds, _ = AO.assign(*(c[k].cpu() if c[k] is not None else None for k in ("mask_logits", "cls_logits", "gt_masks", "gt_labels")))
    assert torch.equal(r.gt_inds.cpu(), ref_inds)
    got = r.get_extra_property("pids").cpu()
    assert torch.equal(got[ref_inds > 0], (99 + ref_inds[ref_inds > 0]))


def test_more_than_127_ground_truth_masks(gpu):
    """an image with more instances than one launch takes columns for: the sums are gathered block-wise"""
    from polyphonicformer_amd import assigner as A
    c = Hh.assign_case(seed=23, N=200, G=150, L=8, H=10, W=16)
    ref_inds, ref_labels = AO.assign(c["mask_logits"], c["cls_logits"], c["gt_masks"], c["gt_labels"], c["gt_valid"])
    ref_cost = AO.cost_matrix(c["mask_logits"], c["cls_logits"], c["gt_masks"], c["gt_labels"], c["gt_valid"])
    d = _dev(c, gpu)
    a = A.build_assigner(dict(CFG))
    sums = A.MatchSums(d["mask_logits"][None], d["gt_masks"][None], d["gt_valid"][None])
    assert Hh.rel_err(a.costs(sums, 0, d["cls_logits"], d["gt_labels"]).cpu(), ref_cost) < 1e-4
    r = a.assign(d["mask_logits"], d["cls_logits"], d["gt_masks"], d["gt_labels"], None, gt_valid=d["gt_valid"])
    assert torch.equal(r.gt_inds.cpu(), ref_inds) and torch.equal(r.labels.cpu(), ref_labels)
