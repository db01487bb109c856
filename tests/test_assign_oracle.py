"""CPU: the assignment oracle (oracle/assign_oracle.py) against goldens written by the reference's own assigner
(oracle/gen_golden_assign.py -> tests/golden/assign.npz), and the host-side logic of the product assigner that needs no GPU."""
import numpy as np
import pytest
import torch

from tests import helpers as Hh
from oracle import assign_oracle as AO


@pytest.fixture(scope="module")
def gold():
    return Hh.load_golden("assign.npz")


@pytest.mark.parametrize("i", range(len(Hh.ASSIGN_CASES)))
def test_oracle_costs_and_assignment_match_reference(gold, i):
    c = Hh.assign_case(**Hh.ASSIGN_CASES[i])
    cost = AO.cost_matrix(c["mask_logits"], c["cls_logits"], c["gt_masks"], c["gt_labels"], c["gt_valid"])
    assert Hh.rel_err(cost, gold[f"c{i}_cost"]) < 1e-6
    inds, labels = AO.assign(c["mask_logits"], c["cls_logits"], c["gt_masks"], c["gt_labels"], c["gt_valid"])
    assert np.array_equal(inds.numpy(), gold[f"c{i}_gt_inds"])
    assert np.array_equal(labels.numpy(), gold[f"c{i}_labels"])


@pytest.mark.parametrize("i", Hh.DEPTH_COST_CASES)
@pytest.mark.parametrize("mode", ["sigmoid", "monodepth"])
def test_oracle_depth_cost_matches_reference(gold, i, mode):
    """DepthCost with weight 0.5 (assigner.py:17-80): the cost matrix and the assignment it leads to"""
    case = Hh.ASSIGN_CASES[i]
    c = Hh.assign_case(**case)
    z, gd = Hh.assign_depth_inputs(case["seed"], case["N"], case["H"], case["W"])
    dc = AO.depth_cost(z, gd, c["gt_masks"], mode, weight=0.5)
    assert Hh.rel_err(dc, gold[f"d{i}_{mode}_depth_cost"]) < 1e-5
    inds, labels = AO.assign(c["mask_logits"], c["cls_logits"], c["gt_masks"], c["gt_labels"], c["gt_valid"], extra_cost=dc)
    assert np.array_equal(inds.numpy(), gold[f"d{i}_{mode}_gt_inds"]) and np.array_equal(labels.numpy(), gold[f"d{i}_{mode}_labels"])


def test_oracle_empty_ground_truth(gold):
    c = Hh.assign_case(seed=16, N=10, G=0, L=8, H=8, W=8)
    inds, labels = AO.assign(c["mask_logits"], c["cls_logits"], c["gt_masks"], c["gt_labels"], c["gt_valid"])
    assert np.array_equal(inds.numpy(), gold["empty_gt_inds"]) and np.array_equal(labels.numpy(), gold["empty_labels"])


def test_product_assigner_registry_and_guards():
    """registry names / kwargs of the shipped config (configs/_base_/models/polyphonic_former.py:170-192); no CPU fallback"""
    from polyphonicformer_amd import assigner as A, _lib
    a = A.build_assigner(dict(type='MaskHungarianAssignerWithDepth', cls_cost=dict(type='FocalLossCost', weight=2.0),
                              dice_cost=dict(type='DiceCost', weight=4.0, pred_act=True),
                              mask_cost=dict(type='MaskCost', weight=1.0, pred_act=True),
                              depth_cost=dict(type='DepthCost', weight=0., loss_fn=dict(type='DepthMatchLoss', loss_weight=1.),
                                              depth_act_mode='sigmoid')))
    assert a.topk == 1 and a.depth_cost.weight == 0
    c = Hh.assign_case(seed=16, N=10, G=0, L=8, H=8, W=8)
    r = a.assign(c["mask_logits"], c["cls_logits"], c["gt_masks"], c["gt_labels"], None, gt_valid=c["gt_valid"])
    assert r.num_gts == 0 and bool((r.gt_inds == 0).all()) and bool((r.labels == -1).all())      # assigner.py:469-475
    c = Hh.assign_case(**Hh.ASSIGN_CASES[2])
    with pytest.raises(_lib.PolyheadError):                                                      # CPU tensors: refuse
        a.assign(c["mask_logits"], c["cls_logits"], c["gt_masks"], c["gt_labels"], None, gt_valid=c["gt_valid"])
    dc = A.build_match_cost(dict(type='DepthCost', weight=1.0))                                  # built; evaluated on the GPU only
    assert dc.loss_fn.eps == 1e-5 and dc.depth_act_mode == 'monodepth'
    # FocalLossCost is host arithmetic: identical to the oracle's restatement
    cls, lab = torch.randn(9, 5), torch.tensor([0, 4, 2])
    assert torch.allclose(A.FocalLossCost(weight=2.0)(cls, lab), AO.focal_cost(cls, lab, 2.0))


@pytest.mark.parametrize("i", range(len(Hh.ASSIGN_CASES)))
def test_mask_pseudo_sampler_matches_reference(gold, i):
    """the sampler that follows the assigner in forward_train (funcs/sampler.py:93-113): index bookkeeping, runs on any
    device; fed the reference's own assignment it must reproduce the reference sampler's record"""
    from polyphonicformer_amd import assigner as A
    c = Hh.assign_case(**Hh.ASSIGN_CASES[i])
    r = A.AssignResult(c["gt_masks"].shape[0], torch.from_numpy(gold[f"c{i}_gt_inds"]), None,
                       labels=torch.from_numpy(gold[f"c{i}_labels"]))
    sr = A.build_sampler(dict(type="MaskPseudoSampler")).sample(r, c["mask_logits"], c["gt_masks"], depth=c["mask_logits"] * 0.5)
    assert np.array_equal(sr.pos_inds.numpy(), gold[f"c{i}_pos_inds"]) and np.array_equal(sr.neg_inds.numpy(), gold[f"c{i}_neg_inds"])
    assert np.array_equal(sr.pos_assigned_gt_inds.numpy(), gold[f"c{i}_pos_assigned_gt_inds"])
    assert np.array_equal(sr.pos_gt_labels.numpy(), gold[f"c{i}_pos_gt_labels"])
    assert np.allclose(sr.pos_gt_masks.sum((1, 2)).numpy(), gold[f"c{i}_pos_gt_masks_sum"], rtol=1e-6)
    assert np.allclose(sr.pos_depth.sum((1, 2)).numpy(), gold[f"c{i}_pos_depth_sum"], rtol=1e-5, atol=1e-4)
    assert sr.num_gts == c["gt_masks"].shape[0] and sr.masks.shape[0] == c["mask_logits"].shape[0]
    assert int(sr.pos_is_gt.sum()) == 0
