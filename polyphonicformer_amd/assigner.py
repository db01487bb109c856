"""SURVEY.md 8f N4 (first part): the mask Hungarian assigner of the training path, with its pixel sums on the device.

Mirrors polyphonic/funcs/assigner.py -- `MaskHungarianAssignerWithDepth` (:363-542, the one the shipped config builds,
configs/_base_/models/polyphonic_former.py:170-192), `MaskHungarianAssigner` (:199-360), `DiceCost` (:80-146),
`MaskCost` (:149-196) -- and mmdet's `FocalLossCost` (mmdet/core/bbox/match_costs/match_cost.py:54-99): same registry
names, constructor kwargs, `assign(...)` signature and `AssignResult` fields.  What changes is where the arithmetic runs:
the three `einsum`s over all pixels and the four row sums of one image (DiceCost.dice_loss :113-129, MaskCost.__call__
:178-194) are ONE pass of `ph_match_sums` (csrc/ph_match.hip: bf16 hi/lo MFMA contraction over the pixel axis with the
sigmoid fused in); the [N, G] cost algebra and `scipy.optimize.linear_sum_assignment` stay on the host, as in the
reference (:511-519).  `DepthCost` (weight 0 in the shipped configs) is one direct pass of `ph_depth_cost_sums` when its weight
is not zero.  No CPU fallback: tensors must live on the GPU."""
import numpy as np
import torch

from . import _lib
from .registry import Registry

try:
    from scipy.optimize import linear_sum_assignment
except ImportError:                                    # the reference raises at assign time (:513-515)
    linear_sum_assignment = None

BBOX_ASSIGNERS = Registry("bbox_assigner")
BBOX_SAMPLERS = Registry("bbox_sampler")
MATCH_COST = Registry("match_cost")


def build_match_cost(cfg):
    return MATCH_COST.build(cfg)


def build_assigner(cfg):
    return BBOX_ASSIGNERS.build(cfg)


def build_sampler(cfg):
    return BBOX_SAMPLERS.build(cfg)


class AssignResult:
    """the fields of mmdet's AssignResult the heads read (mmdet/core/bbox/assigners/assign_result.py:41-48)"""

    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts, self.gt_inds, self.max_overlaps, self.labels = num_gts, gt_inds, max_overlaps, labels
        self._extra_properties = {}

    def set_extra_property(self, key, value):
        self._extra_properties[key] = value

    def get_extra_property(self, key):
        return self._extra_properties.get(key)


class MatchSums:
    """every pixel sum the mask costs need, for a batch of images, from one `ph_match_sums` launch (one per block of 127
    ground-truth masks when an image has more)"""

    GMAX = 127          # ph_match_sums: G + 1 columns <= 128

    def __init__(self, mask_logits, gt_masks, gt_valid=None):
        """mask_logits [B, N, H, W] fp32 cuda; gt_masks [B, G, H, W] (soft masks, zero rows as padding);
        gt_valid [B, H, W] of 0/1 or None"""
        if not mask_logits.is_cuda:
            raise _lib.PolyheadError("MatchSums: tensors must be on the GPU (no CPU fallback in the product path)")
        G = gt_masks.shape[1]
        lg = mask_logits.detach().float().contiguous()
        gt = gt_masks.detach().float()
        va = gt_valid.detach().float().contiguous() if gt_valid is not None else None
        parts = [self._block(lg, gt[:, g0:g0 + self.GMAX].contiguous(), va) for g0 in range(0, max(G, 1), self.GMAX)]
        self.A = torch.cat([p[0] for p in parts], 2)                     # sum p t v            [B, N, G]
        self.S, self.Q, self.V = parts[0][1], parts[0][2], parts[0][5]   # sum p v, sum p^2 v [B, N]; sum v [B]
        self.C = torch.cat([p[3] for p in parts], 1)                     # sum t^2 v            [B, G]
        self.T = torch.cat([p[4] for p in parts], 1)                     # sum t v              [B, G]

    @staticmethod
    def _block(lg, gt, va):
        B, N, H, W = lg.shape
        G = gt.shape[1]
        lib = _lib.load()
        ns, rec = lib.ph_match_nsplit(H * W, B), lib.ph_match_record_floats(N, G)
        part = torch.empty((B, ns, rec), dtype=torch.float32, device=lg.device)
        _lib.check(lib.ph_match_sums(_lib.ptr(lg), _lib.ptr(gt) if G else None, _lib.ptr(va), _lib.ptr(part), B, N, G, H * W,
                                     _lib.stream_ptr()), "ph_match_sums")
        r = part.sum(1)                                                  # fixed order over the splits
        Np, Gp = (N + 31) // 32 * 32, (G + 1 + 31) // 32 * 32      # ph_n_padded
        A = r[:, :Np * Gp].reshape(B, Np, Gp)
        o = Np * Gp
        return (A[:, :N, :G], A[:, :N, G], r[:, o:o + N], r[:, o + Np:o + Np + G], r[:, o + Np + Gp:o + Np + Gp + G],
                r[:, o + Np + 2 * Gp])


@MATCH_COST.register_module()
class FocalLossCost:
    """mmdet/core/bbox/match_costs/match_cost.py:54-99 ([N, L] logits: host-sized arithmetic, plain torch)"""

    def __init__(self, weight=1., alpha=0.25, gamma=2, eps=1e-12):
        self.weight, self.alpha, self.gamma, self.eps = weight, alpha, gamma, eps

    def __call__(self, cls_pred, gt_labels):
        p = cls_pred.sigmoid()
        neg = -(1 - p + self.eps).log() * (1 - self.alpha) * p.pow(self.gamma)
        pos = -(p + self.eps).log() * self.alpha * (1 - p).pow(self.gamma)
        return (pos[:, gt_labels] - neg[:, gt_labels]) * self.weight


@MATCH_COST.register_module()
class DiceCost:
    """assigner.py:80-146.  `from_sums` is the device path; only pred_act with sigmoid is built (shipped configs)."""

    def __init__(self, weight=1., pred_act=False, act_mode='sigmoid', eps=1e-3):
        self.weight, self.pred_act, self.act_mode, self.eps = weight, pred_act, act_mode, eps

    def from_sums(self, s, i):
        d = (2 * s.A[i]) / ((s.Q[i] + self.eps)[:, None] + (s.C[i] + self.eps)[None])          # :125-129
        return -d * self.weight

    def __call__(self, mask_preds, gt_masks, gt_valid=None):
        _need_sigmoid(self)
        return self.from_sums(MatchSums(mask_preds[None], gt_masks[None], None if gt_valid is None else gt_valid[None]), 0)


@MATCH_COST.register_module()
class MaskCost:
    """assigner.py:149-196"""

    def __init__(self, weight=1., pred_act=False, act_mode='sigmoid'):
        self.weight, self.pred_act, self.act_mode = weight, pred_act, act_mode

    def from_sums(self, s, i):
        pos = s.A[i]                                                                            # :185 / :192
        neg = s.V[i] - s.S[i][:, None] - s.T[i][None] + s.A[i]        # sum (1 - p)(1 - t) v           :186 / :193
        return -(pos + neg) / s.V[i] * self.weight                                              # :190 / :194

    def __call__(self, cls_pred, target, gt_valid=None):
        _need_sigmoid(self)
        return self.from_sums(MatchSums(cls_pred[None], target[None], None if gt_valid is None else gt_valid[None]), 0)


@MATCH_COST.register_module()
class DepthMatchLoss:
    """assigner.py:17-46: loss_weight * (loss_si * si + loss_sq_rel * sq_rel + loss_abs_rel * abs_rel) from the pixel sums"""

    def __init__(self, loss_weight=1., loss_si=1., loss_sq_rel=1., loss_abs_rel=1.):
        self.loss_weight, self.loss_si, self.loss_sq_rel, self.loss_abs_rel = loss_weight, loss_si, loss_sq_rel, loss_abs_rel
        self.eps = 1.e-5

    def from_sums(self, s, num_valid):
        """s [N, G, 4] = sum lm^2, sum lm, sum r^2, sum |r|; num_valid [G] already clamped"""
        si = s[..., 0] / num_valid - s[..., 1] / torch.square(num_valid)
        sq = torch.sqrt(s[..., 2] / num_valid)
        ab = s[..., 3] / num_valid
        return self.loss_weight * (self.loss_si * si + self.loss_sq_rel * sq + self.loss_abs_rel * ab)


@MATCH_COST.register_module()
class DepthCost:
    """assigner.py:49-80: the DepthMatchLoss of every (predicted depth map, ground-truth instance) pair over the instance's
    pixels with a depth label -- one pass of `ph_depth_cost_sums` (csrc/ph_match.hip)"""

    def __init__(self, weight=1., loss_fn=dict(type='DepthMatchLoss', loss_weight=1.), depth_act_mode='monodepth'):
        self.weight = weight
        self.loss_fn = build_match_cost(dict(loss_fn))
        self.depth_act_mode = depth_act_mode

    def __call__(self, inputs, depth_gt, target_masks):
        if not inputs.is_cuda:
            raise _lib.PolyheadError("DepthCost: tensors must live on the GPU (libpolyhead has no CPU path)")
        n, m = inputs.shape[0], target_masks.shape[0]
        HW = inputs[0].numel()
        z = inputs.detach().contiguous().float()
        gd = depth_gt.detach().to(z.device).contiguous().float().reshape(-1)
        tm = target_masks.detach().contiguous().float()
        assert gd.numel() == HW and tm[0].numel() == HW
        out = torch.empty((n, m, 4), dtype=torch.float32, device=z.device)
        nv = torch.empty((m,), dtype=torch.float32, device=z.device)
        _lib.check(_lib.load().ph_depth_cost_sums(_lib.ptr(z), _lib.ptr(gd), _lib.ptr(tm), n, m, HW,
                                                  {"sigmoid": 0, "monodepth": 1}[self.depth_act_mode], self.loss_fn.eps,
                                                  _lib.ptr(out), _lib.ptr(nv), _lib.stream_ptr()), "ph_depth_cost_sums")
        return self.loss_fn.from_sums(out, nv.clamp(min=0.001)[None]) * self.weight       # :77-80


def _need_sigmoid(c):
    if not (c.pred_act and c.act_mode == 'sigmoid'):
        raise NotImplementedError("only pred_act=True with act_mode='sigmoid' (the shipped configs) runs on the device")


def _hungarian(cost, topk):
    """assigner.py:509-531"""
    cost = cost.detach().cpu()
    if linear_sum_assignment is None:
        raise ImportError('Please run "pip install scipy" to install scipy first.')
    if topk == 1:
        return linear_sum_assignment(cost)
    rows, cols = [], []
    for _ in range(topk):
        r, c = linear_sum_assignment(cost)
        rows.append(r)
        cols.append(c)
        cost[r] = 1e10
    return np.concatenate(rows), np.concatenate(cols)


class _MaskAssignerBase:
    def __init__(self, cls_cost=dict(type='ClassificationCost', weight=1.), mask_cost=dict(type='SigmoidCost', weight=1.0),
                 dice_cost=dict(), depth_cost=None, boundary_cost=None, topk=1):
        self.cls_cost = build_match_cost(cls_cost)
        self.mask_cost = build_match_cost(mask_cost)
        self.dice_cost = build_match_cost(dice_cost)
        if boundary_cost is not None:
            raise NotImplementedError("boundary_cost is not used by the shipped configs")
        self.boundary_cost = None
        self.depth_cost = build_match_cost(depth_cost) if depth_cost is not None else None
        self.topk = topk
        for c in (self.mask_cost, self.dice_cost):
            if c.weight != 0:
                _need_sigmoid(c)

    def costs(self, sums, i, cls_pred, gt_labels, depth_pred=None, gt_depth=None, gt_masks=None):
        """weighted cost matrix [N, G] of image i of a `MatchSums` batch (assigner.py:478-506)"""
        cost = 0
        if self.depth_cost is not None and self.depth_cost.weight != 0 and gt_depth is not None and depth_pred is not None:
            cost = cost + self.depth_cost(inputs=depth_pred, depth_gt=gt_depth, target_masks=gt_masks)           # :497-502
        if self.cls_cost.weight != 0 and cls_pred is not None:
            cost = cost + self.cls_cost(cls_pred, gt_labels)
        if self.mask_cost.weight != 0:
            cost = cost + self.mask_cost.from_sums(sums, i)
        if self.dice_cost.weight != 0:
            cost = cost + self.dice_cost.from_sums(sums, i)
        return cost

    def _assign(self, bbox_pred, cls_pred, gt_bboxes, gt_labels, gt_valid, gt_pids=None, depth_pred=None, gt_depth=None):
        num_gts, num_bboxes = gt_bboxes.size(0), bbox_pred.size(0)
        gt_inds = bbox_pred.new_full((num_bboxes,), -1, dtype=torch.long)                      # :463-468
        labels = bbox_pred.new_full((num_bboxes,), -1, dtype=torch.long)
        if num_gts == 0 or num_bboxes == 0:                                                    # :469-475
            if num_gts == 0:
                gt_inds[:] = 0
            return AssignResult(num_gts, gt_inds, None, labels=labels)
        sums = MatchSums(bbox_pred[None], gt_bboxes[None], None if gt_valid is None else gt_valid[None])
        cost = self.costs(sums, 0, cls_pred, gt_labels, depth_pred, gt_depth, gt_bboxes)
        rows, cols = _hungarian(cost, self.topk)
        rows = torch.from_numpy(rows).to(bbox_pred.device)
        cols = torch.from_numpy(cols).to(bbox_pred.device)
        gt_inds[:] = 0                                                                         # :535-540
        gt_inds[rows] = cols + 1
        labels[rows] = gt_labels[cols]
        res = AssignResult(num_gts, gt_inds, None, labels=labels)
        if gt_pids is not None:                                                                # :343-349
            pids = bbox_pred.new_full((num_bboxes,), -1, dtype=torch.long)
            pids[rows] = gt_pids[cols]
            res.set_extra_property("pids", pids)
        return res


@BBOX_ASSIGNERS.register_module()
class MaskHungarianAssignerWithDepth(_MaskAssignerBase):
    """assigner.py:363-542"""

    def assign(self, bbox_pred, cls_pred, gt_bboxes, gt_labels, img_meta=None, gt_bboxes_ignore=None, depth_pred=None,
               gt_depth=None, gt_valid=None, eps=1e-7):
        assert gt_bboxes_ignore is None, 'Only case when gt_bboxes_ignore is None is supported.'
        return self._assign(bbox_pred, cls_pred, gt_bboxes, gt_labels, gt_valid, depth_pred=depth_pred, gt_depth=gt_depth)


@BBOX_ASSIGNERS.register_module()
class MaskHungarianAssigner(_MaskAssignerBase):
    """assigner.py:199-360 (no gt_valid, optional gt_pids)"""

    def __init__(self, cls_cost=dict(type='ClassificationCost', weight=1.), mask_cost=dict(type='SigmoidCost', weight=1.0),
                 dice_cost=dict(), boundary_cost=None, topk=1):
        super().__init__(cls_cost, mask_cost, dice_cost, None, boundary_cost, topk)

    def assign(self, bbox_pred, cls_pred, gt_bboxes, gt_labels, gt_pids=None, img_meta=None, gt_bboxes_ignore=None, eps=1e-7):
        assert gt_bboxes_ignore is None, 'Only case when gt_bboxes_ignore is None is supported.'
        return self._assign(bbox_pred, cls_pred, gt_bboxes, gt_labels, None, gt_pids)


class MaskSamplingResult:
    """the record `get_targets` reads (kernel_update_head.py:443-590), field for field what funcs/sampler.py:26-51 builds:
    positives / negatives of one image, their predictions, and for the positives the matched ground truth.  Index
    bookkeeping only."""

    def __init__(self, pos_inds, neg_inds, masks, gt_masks, assign_result, gt_flags, depth=None):
        pick = lambda t, idx: None if t is None else t[idx]
        self.pos_inds, self.neg_inds = pos_inds, neg_inds
        self.pos_masks, self.neg_masks = pick(masks, pos_inds), pick(masks, neg_inds)
        self.pos_depth, self.neg_depth = pick(depth, pos_inds), pick(depth, neg_inds)
        self.pos_is_gt = pick(gt_flags, pos_inds)
        self.num_gts = int(gt_masks.shape[0])
        self.pos_assigned_gt_inds = pick(assign_result.gt_inds, pos_inds) - 1      # gt_inds are 1-based, 0 = background
        if gt_masks.numel():
            self.pos_gt_masks = gt_masks[self.pos_assigned_gt_inds]
        else:                                                                       # no ground truth: nothing can be positive
            if self.pos_assigned_gt_inds.numel():
                raise AssertionError("positives without ground truth")
            self.pos_gt_masks = torch.empty_like(gt_masks)
        self.pos_gt_labels = pick(assign_result.labels, pos_inds)

    @property
    def masks(self):
        """positives first, then negatives"""
        return torch.cat((self.pos_masks, self.neg_masks), 0)

    @property
    def info(self):
        keys = ("pos_inds", "neg_inds", "pos_masks", "neg_masks", "pos_is_gt", "num_gts", "pos_assigned_gt_inds")
        return {k: getattr(self, k) for k in keys}


@BBOX_SAMPLERS.register_module()
class MaskPseudoSampler:
    """funcs/sampler.py:81-113 -- no sampling: every matched prediction is a positive, every background one a negative"""

    def __init__(self, **kwargs):
        pass

    def sample(self, assign_result, masks, gt_masks, depth=None, **kwargs):
        gi = assign_result.gt_inds
        pos = (gi > 0).nonzero().flatten()             # ascending and unique by construction
        neg = (gi == 0).nonzero().flatten()
        flags = torch.zeros(masks.shape[0], dtype=torch.uint8, device=masks.device)
        return MaskSamplingResult(pos, neg, masks, gt_masks, assign_result, flags, depth)
