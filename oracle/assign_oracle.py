"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the mask Hungarian assignment (first part).

Follows polyphonic/funcs/assigner.py: DiceCost.dice_loss :113-129 (pred_act sigmoid :142-146), MaskCost.__call__ :178-196,
MaskHungarianAssignerWithDepth.assign :478-541, and mmdet's FocalLossCost (mmdet/core/bbox/match_costs/match_cost.py:93-99).
Pinned: tests/test_assign_oracle.py checks it against tests/golden/assign.npz, which oracle/gen_golden_assign.py wrote by
running the reference classes themselves.  Only tests / smoke / bench's cpu leg may import this module."""
import numpy as np
import torch
from scipy.optimize import linear_sum_assignment


def focal_cost(cls_pred, gt_labels, weight=2.0, alpha=0.25, gamma=2, eps=1e-12):
    p = cls_pred.sigmoid()                                                        # match_cost.py:93
    neg = -(1 - p + eps).log() * (1 - alpha) * p.pow(gamma)                       # :94-95
    pos = -(p + eps).log() * alpha * (1 - p).pow(gamma)                           # :96-97
    return (pos[:, gt_labels] - neg[:, gt_labels]) * weight                       # :98-99


def dice_cost(mask_logits, gt_masks, gt_valid=None, weight=4.0, eps=1e-3):
    x = mask_logits.sigmoid().reshape(mask_logits.shape[0], -1)                   # assigner.py:142-143, :114
    t = gt_masks.reshape(gt_masks.shape[0], -1).float()                           # :115
    v = torch.ones(x.shape[1]) if gt_valid is None else gt_valid.reshape(-1)      # :116-117 / :123-127
    a = torch.einsum('nh,mh,h->nm', x, t, v)                                      # :120
    b = torch.sum(x * x * v, 1) + eps                                             # :121
    c = torch.sum(t * t * v, 1) + eps                                             # :122
    return -(2 * a) / (b[:, None] + c[None]) * weight                             # :128-130, :147


def mask_cost(mask_logits, gt_masks, gt_valid=None, weight=1.0):
    p = mask_logits.sigmoid()                                                     # assigner.py:174-175
    H, W = gt_masks.shape[-2:]
    if gt_valid is not None:
        pos = torch.einsum('nhw,mhw,hw->nm', p, gt_masks, gt_valid)               # :185
        neg = torch.einsum('nhw,mhw,hw->nm', 1 - p, 1 - gt_masks, gt_valid)       # :186
        c = -(pos + neg) / gt_valid.sum()                                         # :190
    else:
        pos = torch.einsum('nhw,mhw->nm', p, gt_masks)                            # :192
        neg = torch.einsum('nhw,mhw->nm', 1 - p, 1 - gt_masks)                    # :193
        c = -(pos + neg) / (H * W)                                                # :194
    return c * weight


def depth_cost(depth_logits, gt_depth, gt_masks, mode="sigmoid", weight=1.0, loss_weight=1.0, eps=1e-5):
    """DepthCost.__call__ + DepthMatchLoss.__call__ (assigner.py:17-80): every (prediction, ground truth) pair over the
    pixels where gt_depth * gt_mask > 0; the eps-shifted zeros outside them cancel."""
    from .poly_oracle import depth_act
    d = depth_act(depth_logits, mode)                                             # [n, H, W]   :68
    t = gt_depth.reshape(1, *gt_masks.shape[-2:]) * gt_masks                      # [m, H, W]   :70
    v = (t > 0).float()                                                           # :75
    di = d[:, None] * v[None] + eps                                               # :76, :33
    ti = (t[None] + eps).expand_as(di)                                            # :34
    nv = v.sum((-1, -2)).clamp(min=0.001)[None]                                   # :77
    lm, mi = torch.log(di) - torch.log(ti), di - ti                               # :35-36
    si = (lm ** 2).sum((-1, -2)) / nv - lm.sum((-1, -2)) / nv ** 2                # :38-39
    sq = torch.sqrt(((mi / ti) ** 2).sum((-1, -2)) / nv)                          # :41
    ab = (mi / ti).abs().sum((-1, -2)) / nv                                       # :43
    return loss_weight * (si + sq + ab) * weight                                  # :45-46, :80


def cost_matrix(mask_logits, cls_logits, gt_masks, gt_labels, gt_valid=None, w_cls=2.0, w_dice=4.0, w_mask=1.0):
    c = dice_cost(mask_logits, gt_masks, gt_valid, w_dice) + mask_cost(mask_logits, gt_masks, gt_valid, w_mask)
    if cls_logits is not None:
        c = c + focal_cost(cls_logits, gt_labels, w_cls)
    return c                                                                      # :506


def assign(mask_logits, cls_logits, gt_masks, gt_labels, gt_valid=None, extra_cost=None, **w):
    """-> (assigned_gt_inds [N] (0 = background, k = gt k-1), assigned_labels [N] (-1 = none))   assigner.py:463-541"""
    N, G = mask_logits.shape[0], gt_masks.shape[0]
    inds = torch.full((N,), -1, dtype=torch.long)
    labels = torch.full((N,), -1, dtype=torch.long)
    if G == 0 or N == 0:
        if G == 0:
            inds[:] = 0
        return inds, labels
    cost = cost_matrix(mask_logits, cls_logits, gt_masks, gt_labels, gt_valid, **w)
    if extra_cost is not None:                                                    # the depth cost (:497-504)
        cost = cost + extra_cost
    rows, cols = linear_sum_assignment(cost.numpy())
    inds[:] = 0
    inds[torch.from_numpy(rows)] = torch.from_numpy(cols) + 1
    labels[torch.from_numpy(rows)] = gt_labels[torch.from_numpy(cols)]
    return inds, labels
