"""TEST INFRASTRUCTURE ONLY -- writes tests/golden/assign.npz by running the REFERENCE assigner
(polyphonic/funcs/assigner.py MaskHungarianAssignerWithDepth with the shipped config's costs,
configs/_base_/models/polyphonic_former.py:178-192) on the seeded cases of tests/helpers.py (ASSIGN_CASES).
Run in the build container:  PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_assign.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_loader as R          # noqa: E402
from tests import helpers as Hh            # noqa: E402

CFG = dict(cls_cost=dict(type='FocalLossCost', weight=2.0), dice_cost=dict(type='DiceCost', weight=4.0, pred_act=True),
           mask_cost=dict(type='MaskCost', weight=1.0, pred_act=True),
           depth_cost=dict(type='DepthCost', weight=0., loss_fn=dict(type='DepthMatchLoss', loss_weight=1.),
                           depth_act_mode='sigmoid'))


def main():
    ns = R.load_reference_assigner()
    a = ns.Assigner(**CFG)
    sampler = sys.modules["polyphonic.funcs.sampler"].MaskPseudoSampler()
    out = {}
    for i, case in enumerate(Hh.ASSIGN_CASES):
        c = Hh.assign_case(**case)
        r = a.assign(c["mask_logits"], c["cls_logits"], c["gt_masks"], c["gt_labels"], None, gt_valid=c["gt_valid"])
        cost = a.mask_cost(c["mask_logits"], c["gt_masks"], c["gt_valid"]) + a.dice_cost(c["mask_logits"], c["gt_masks"], c["gt_valid"])
        if c["cls_logits"] is not None:
            cost = cost + a.cls_cost(c["cls_logits"], c["gt_labels"])
        out[f"c{i}_cost"] = cost.numpy().astype(np.float32)
        out[f"c{i}_gt_inds"] = r.gt_inds.numpy()
        out[f"c{i}_labels"] = r.labels.numpy()
        # the sampler that follows the assigner in forward_train (kernel_update.py:247-251, funcs/sampler.py:93-113)
        sr = sampler.sample(r, c["mask_logits"], c["gt_masks"], depth=c["mask_logits"] * 0.5)
        out[f"c{i}_pos_inds"], out[f"c{i}_neg_inds"] = sr.pos_inds.numpy(), sr.neg_inds.numpy()
        out[f"c{i}_pos_assigned_gt_inds"], out[f"c{i}_pos_gt_labels"] = sr.pos_assigned_gt_inds.numpy(), sr.pos_gt_labels.numpy()
        out[f"c{i}_pos_gt_masks_sum"] = sr.pos_gt_masks.sum((1, 2)).numpy().astype(np.float32)
        out[f"c{i}_pos_depth_sum"] = sr.pos_depth.sum((1, 2)).numpy().astype(np.float32)
        print(i, case, "matched", int((r.gt_inds > 0).sum()))
    # DepthCost with a non-zero weight (assigner.py:17-80, :497-502), both depth activations
    for i in Hh.DEPTH_COST_CASES:
        case = Hh.ASSIGN_CASES[i]
        c = Hh.assign_case(**case)
        z, gd = Hh.assign_depth_inputs(case["seed"], case["N"], case["H"], case["W"])
        for mode in ("sigmoid", "monodepth"):
            cfg = dict(CFG, depth_cost=dict(type='DepthCost', weight=0.5, loss_fn=dict(type='DepthMatchLoss', loss_weight=1.),
                                            depth_act_mode=mode))
            ad = ns.Assigner(**cfg)
            r = ad.assign(c["mask_logits"], c["cls_logits"], c["gt_masks"], c["gt_labels"], None, depth_pred=z, gt_depth=gd,
                          gt_valid=c["gt_valid"])
            out[f"d{i}_{mode}_depth_cost"] = ad.depth_cost(inputs=z, depth_gt=gd, target_masks=c["gt_masks"]).numpy().astype(np.float32)
            out[f"d{i}_{mode}_gt_inds"], out[f"d{i}_{mode}_labels"] = r.gt_inds.numpy(), r.labels.numpy()
            print("depth cost", i, mode, "matched", int((r.gt_inds > 0).sum()),
                  "differs from the depth-free assignment at", int((r.gt_inds.numpy() != out[f"c{i}_gt_inds"]).sum()), "rows")
    # empty ground truth (assigner.py:469-475)
    c = Hh.assign_case(seed=16, N=10, G=0, L=8, H=8, W=8)
    r = a.assign(c["mask_logits"], c["cls_logits"], c["gt_masks"], c["gt_labels"], None, gt_valid=c["gt_valid"])
    out["empty_gt_inds"], out["empty_labels"] = r.gt_inds.numpy(), r.labels.numpy()
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "assign.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
