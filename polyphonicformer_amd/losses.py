"""Training-side losses of the hot path on libpolyhead's loss kernels (csrc/ph_loss.hip).

Registry names, constructor kwargs and attributes are the reference's: `FocalLoss`, `CrossEntropyLoss`, `DiceLoss`
(vendored mmdet, mmdet/models/losses/*.py) and `DepthLoss` (polyphonic/losses/depth_loss.py).  The modules hold the
configuration; `stage_losses` is what `KernelUpdateHead.loss` (kernel_update_head.py:355-441) runs: four `*_sums` passes
(mask BCE + dice, rank, depth, focal), a handful of scalars combined on the device in fp64 in a fixed order, and -- when
asked -- four `*_grad` passes that write d(sum of the stage's losses) / d(mask_pred, cls_score, depth_pred), the first
step of the backward pass.  The reference does the same arithmetic as ~40 ATen launches with boolean-mask gathers.
These functions evaluate values and d(losses)/d(predictions) without an autograd graph of their own; `train.py` wraps them
into one autograd node per head and stage (`_Objective`), behind which the hand-written backward of the forward kernels runs."""
import torch
import torch.nn as nn

from . import _lib
from .registry import LOSSES

DEPTH_MODES = {"sigmoid": 0, "monodepth": 1}


def _gpu(t, name):
    if not t.is_cuda:
        raise _lib.PolyheadError(f"{name} must live on the GPU: libpolyhead has no CPU path")


def _f32(t):
    return t.detach().contiguous().float()


# ---- the four passes --------------------------------------------------------------------------------------------------------
def mask_loss_sums(pred, target, weight, pos_rows, nsplit=None):
    """pred / target / weight [R, HW] fp32, pos_rows int32 [P] -> float64 [P, 5] (BCE sum, count, a, b, c), fixed order"""
    P, HW = pos_rows.numel(), pred.shape[1]
    nsplit = nsplit or max(1, min(64, 2048 // max(P, 1)))
    out = torch.empty((P, nsplit, 5), dtype=torch.float64, device=pred.device)
    lib = _lib.load()
    _lib.check(lib.ph_mask_loss_sums(_lib.ptr(pred), _lib.ptr(target), _lib.ptr(weight), _lib.ptr(pos_rows), P, HW, nsplit,
                                     _lib.ptr(out), _lib.stream_ptr()), "ph_mask_loss_sums")
    return out.sum(1)


def rank_loss_sum(pred, rank_target, ignore):
    B, N, H, W = pred.shape
    lib = _lib.load()
    out = torch.empty((B * lib.ph_rank_loss_blocks(H * W),), dtype=torch.float64, device=pred.device)
    _lib.check(lib.ph_rank_loss_sum(_lib.ptr(pred), _lib.ptr(rank_target), B, N, H * W, ignore, _lib.ptr(out), _lib.stream_ptr()),
               "ph_rank_loss_sum")
    return out.sum()


def depth_loss_sums(pred, target, weight, mode):
    lib = _lib.load()
    total = pred.numel()
    out = torch.empty((lib.ph_depth_loss_blocks(total), 5), dtype=torch.float64, device=pred.device)
    _lib.check(lib.ph_depth_loss_sums(_lib.ptr(pred), _lib.ptr(target), _lib.ptr(weight), total, mode, _lib.ptr(out),
                                      _lib.stream_ptr()), "ph_depth_loss_sums")
    return out.sum(0)


def seg_focal_sum(pred, target, gamma, alpha):
    """pred [B, L, HW] fp32 (class-major, as conv_seg writes it), target int32 [B, HW] (== L: pixel not selected)"""
    lib = _lib.load()
    B, L, HW = pred.shape
    nb = lib.ph_rank_loss_blocks(HW)
    out = torch.empty((B * nb,), dtype=torch.float64, device=pred.device)
    _lib.check(lib.ph_seg_focal_sum(_lib.ptr(pred), _lib.ptr(target), B, L, HW, gamma, alpha, _lib.ptr(out), _lib.stream_ptr()),
               "ph_seg_focal_sum")
    return out.sum()


def focal_loss_sum(pred, labels, weight, gamma, alpha):
    lib = _lib.load()
    R, L = pred.shape
    out = torch.empty((lib.ph_focal_loss_blocks(R * L),), dtype=torch.float64, device=pred.device)
    _lib.check(lib.ph_focal_loss_sum(_lib.ptr(pred), _lib.ptr(labels), _lib.ptr(weight), R, L, gamma, alpha, _lib.ptr(out),
                                     _lib.stream_ptr()), "ph_focal_loss_sum")
    return out.sum()


# ---- loss modules (configuration holders with the reference's names) -----------------------------------------------------------
class _Loss(nn.Module):
    def __init__(self, use_sigmoid=False, loss_weight=1.0, reduction="mean", **kw):
        super().__init__()
        self.use_sigmoid, self.loss_weight, self.reduction = use_sigmoid, loss_weight, reduction
        self.cfg = dict(kw)


# the process group the focal normaliser is averaged over (None: the default group); train.TrainStep(group=...) sets it for
# the duration of its step so that a sub-group step neither averages over foreign ranks nor waits for them
_REDUCE_GROUP = [None]


class reduce_group:
    def __init__(self, group):
        self.group = group

    def __enter__(self):
        _REDUCE_GROUP.append(self.group)

    def __exit__(self, *exc):
        _REDUCE_GROUP.pop()
        return False


class FocalLoss(_Loss):
    """mmdet FocalLoss (focal_loss.py:160-244), the sigmoid form: forward(pred [R, L], target [R], weight [R, L], avg_factor)"""

    def __init__(self, use_sigmoid=True, gamma=2.0, alpha=0.25, reduction="mean", loss_weight=1.0, activated=False, **kw):
        super().__init__(use_sigmoid, loss_weight, reduction, **kw)
        if not use_sigmoid or activated:
            raise NotImplementedError("libpolyhead: sigmoid focal loss on logits only (the shipped configs)")
        self.gamma, self.alpha = gamma, alpha

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        _gpu(pred, "pred")
        pred = _f32(pred)
        red = reduction_override or self.reduction
        if red not in ("mean", "sum") or (red == "sum" and avg_factor is not None):
            raise NotImplementedError("libpolyhead FocalLoss: reduction 'mean' (optionally with avg_factor) or 'sum'")
        if weight is None:
            w = torch.ones_like(pred)
        else:
            w = _f32(weight)
            if w.dim() == 1:                   # one weight per row, as mmdet reshapes it (focal_loss.py:60-69)
                w = w.view(-1, 1)
            w = w.expand_as(pred).contiguous()
        s = focal_loss_sum(pred, target.long().contiguous(), w, self.gamma, self.alpha)
        if avg_factor is not None:
            s = s / avg_factor
        elif red == "mean":
            s = s / pred.numel()
        return (self.loss_weight * s).float()


class CrossEntropyLoss(_Loss):
    """mmdet CrossEntropyLoss (cross_entropy_loss.py:163-251): the stage uses the sigmoid form on selected pixels (loss_mask)
    and the softmax form over the mask channels (loss_rank); both are evaluated inside `stage_losses`"""

    def __init__(self, use_sigmoid=False, use_mask=False, reduction="mean", class_weight=None, ignore_index=None,
                 loss_weight=1.0, **kw):
        super().__init__(use_sigmoid, loss_weight, reduction, **kw)
        if use_mask or class_weight is not None:
            raise NotImplementedError("libpolyhead: use_mask / class_weight are not part of the shipped configs")
        self.ignore_index = ignore_index

    def forward(self, cls_score, label, weight=None, avg_factor=None, reduction_override=None, ignore_index=None, **kw):
        """mmdet CrossEntropyLoss.forward on the device's loss kernels, the two forms the path uses:
        use_sigmoid: binary cross entropy of mask logits [R, P] against soft targets [R, P] (every pixel selected) --
          mean over all elements (cross_entropy_loss.py:56-105), what `loss_mask` evaluates;
        softmax: cross entropy over dim 1 of [B, N, H, W] (or [B, N, P]) logits against an index map [B, H, W] with
          `ignore_index` pixels contributing zero -- mean over all pixels (:9-53), what `loss_rank` evaluates.
        The heads themselves go through `stage_losses` / `rpn_losses`, which fuse these with the other losses."""
        red = reduction_override or self.reduction
        if weight is not None or avg_factor is not None or red != "mean":
            raise NotImplementedError("libpolyhead CrossEntropyLoss: mean reduction without weights (the shipped configs)")
        _gpu(cls_score, "cls_score")
        if self.use_sigmoid:
            z, t = _f32(cls_score).reshape(cls_score.shape[0], -1), _f32(label).reshape(cls_score.shape[0], -1)
            sel = torch.ones_like(z)
            sums = mask_loss_sums(z, t, sel, torch.arange(z.shape[0], device=z.device, dtype=torch.int32))
            return (self.loss_weight * sums[:, 0].sum() / z.numel()).float()
        ii = self.ignore_index if ignore_index is None else ignore_index
        z = _f32(cls_score)
        B, N = z.shape[:2]
        HW = z[0, 0].numel()
        tgt = label.reshape(B, HW).to(torch.int32).contiguous()
        s = rank_loss_sum(z.reshape(B, N, HW, 1), tgt, -100 if ii is None else int(ii))
        return (self.loss_weight * s / (B * HW)).float()


class DiceLoss(_Loss):
    """mmdet DiceLoss (dice_loss.py:49-136), sigmoid activation inside, eps 1e-3"""

    def __init__(self, use_sigmoid=True, activate=True, reduction="mean", loss_weight=1.0, eps=1e-3, **kw):
        super().__init__(use_sigmoid, loss_weight, reduction, **kw)
        if not (use_sigmoid and activate):
            raise NotImplementedError("libpolyhead: DiceLoss with the sigmoid inside (the shipped configs)")
        self.eps, self.activate = eps, activate

    def forward(self, pred, target, weight=None, reduction_override=None, avg_factor=None):
        """mmdet DiceLoss.forward (dice_loss.py:7-47,96-136): pred [R, P] logits, target [R, P]; per-row
        1 - 2 sum(p t) / ((sum(p^2) + eps) + (sum(t^2) + eps)) with p = sigmoid(pred), mean over the rows"""
        red = reduction_override or self.reduction
        if weight is not None or avg_factor is not None or red != "mean":
            raise NotImplementedError("libpolyhead DiceLoss: mean reduction without weights (the shipped configs)")
        _gpu(pred, "pred")
        z, t = _f32(pred).reshape(pred.shape[0], -1), _f32(target).reshape(pred.shape[0], -1)
        sums = mask_loss_sums(z, t, torch.ones_like(z), torch.arange(z.shape[0], device=z.device, dtype=torch.int32))
        dice = 1.0 - 2.0 * sums[:, 2] / (sums[:, 3] + sums[:, 4] + 2.0 * self.eps)          # dice_loss.py:33-38
        return (self.loss_weight * dice.mean()).float()


def _depth_from_sums(s, loss_weight, w3):
    """DepthLoss from the five sums (depth_loss.py:19-32): returns (loss fp64 scalar tensor, coefficients of the grad pass)"""
    n = s[0]
    if float(n) == 0.0:
        return s[0] * 0.0, (0.0, 0.0, 0.0, 0.0)
    si = s[1] / n - s[2] / (n * n)                      # as the reference writes it: sum(log_minus) / n^2, not its square
    sq = torch.sqrt(s[3] / n)
    ab = s[4] / n
    K = loss_weight / 3.0
    loss = K * (w3[0] * si + w3[1] * sq + w3[2] * ab)
    nf, sqf = float(n), float(sq)
    coef = (K * w3[0] * 2.0 / nf, -K * w3[0] / (nf * nf), (K * w3[1] / (nf * sqf)) if sqf > 0 else 0.0, K * w3[2] / nf)
    return loss, coef


class DepthLoss(_Loss):
    """polyphonic/losses/depth_loss.py:36-65: forward(pred logits, target, mask_weight) -> loss_weight * mean(weight * (si, sq_rel, abs_rel))"""

    def __init__(self, loss_weight=1.0, depth_act_mode="monodepth", si_weight=1.0, sq_rel_weight=1.0, abs_rel_weight=1.0, **kw):
        super().__init__(False, loss_weight, "mean", **kw)
        self.depth_act_mode = depth_act_mode
        self.weight = torch.tensor([si_weight, sq_rel_weight, abs_rel_weight], dtype=torch.float32)

    def forward(self, pred, target, mask_weight, reduction_override="mean", **kwargs):
        if not torch.any(self.weight > 0):
            return (pred * 0).sum()
        _gpu(pred, "pred")
        s = depth_loss_sums(_f32(pred), _f32(target), _f32(mask_weight), DEPTH_MODES[self.depth_act_mode])
        loss, _ = _depth_from_sums(s, self.loss_weight, [float(v) for v in self.weight])
        if reduction_override == "sum":
            loss = loss * 3.0
        return loss.float()


for _n, _c in (("FocalLoss", FocalLoss), ("CrossEntropyLoss", CrossEntropyLoss), ("DiceLoss", DiceLoss), ("DepthLoss", DepthLoss)):
    LOSSES.register_module(name=_n, module=_c, force=True)


def _mask_terms(head, mp, mask_targets, mask_weights, pos, B, N, H, W, losses, grad, keys, empty_keys):
    """The mask part both heads share (kernel_update_head.py:404-441, kernel_head.py:503-536): BCE mean over the selected
    pixels of the positive rows, the mean of the per-row dice losses, the rank cross-entropy.  mp [R, HW] fp32; `grad` (or
    None) [R, HW] receives d(sum of the three) / d(mask_pred).  `keys` / `empty_keys` name the entries with / without
    positives (the reference's names differ between the two cases)."""
    lib, dev = _lib.load(), mp.device
    R, HW = B * N, H * W
    ignore, lr = head.ignore_label, head.loss_rank
    num_pos = int(pos.sum())
    if not num_pos:
        z = torch.zeros((), device=dev)
        losses[empty_keys[0]], losses[empty_keys[1]] = z, z.clone()
        if lr is not None:
            losses[empty_keys[2]] = z.clone()
        if grad is not None:
            grad.zero_()
        return
    mt, mw = _f32(mask_targets).reshape(R, HW), _f32(mask_weights).reshape(R, HW)
    rows = pos.nonzero().flatten().to(torch.int32).contiguous()
    s = mask_loss_sums(mp, mt, mw, rows)                                                       # [P, 5] fp64
    ntot = s[:, 1].sum()
    lm, ldice = head.loss_mask, head.loss_dice
    losses[keys[0]] = (lm.loss_weight * s[:, 0].sum() / ntot).float()                          # BCE mean over the selected pixels
    bc = s[:, 3] + s[:, 4] + 2 * ldice.eps
    losses[keys[1]] = (ldice.loss_weight * (1 - 2 * s[:, 2] / bc).mean()).float()
    rank_target = None
    if lr is not None:
        # rank target: pixel -> index (within its image) of the LAST positive row whose target covers it
        rank_target = torch.empty((B, HW), dtype=torch.int32, device=dev)
        pos_u8 = pos.to(torch.uint8).contiguous()      # named: alive until the launch is queued
        _lib.check(lib.ph_rank_target(_lib.ptr(mt), _lib.ptr(pos_u8), B, N, HW, ignore, _lib.ptr(rank_target),
                                      _lib.stream_ptr()), "ph_rank_target")
        losses[keys[2]] = (lr.loss_weight * rank_loss_sum(mp.reshape(B, N, H, W), rank_target, ignore) / (B * HW)).float()
    if grad is not None:
        if lr is not None:
            _lib.check(lib.ph_rank_loss_grad(_lib.ptr(mp), _lib.ptr(rank_target), B, N, HW, ignore, lr.loss_weight / (B * HW),
                                             _lib.ptr(grad), _lib.stream_ptr()), "ph_rank_loss_grad")
        else:
            grad.zero_()
        P = rows.numel()
        coef = torch.stack([torch.full((P,), lm.loss_weight, dtype=torch.float64, device=dev) / ntot,
                            -2.0 * ldice.loss_weight / (P * bc), 4.0 * ldice.loss_weight * s[:, 2] / (P * bc * bc)], 1)
        coef = coef.float().contiguous()
        _lib.check(lib.ph_mask_loss_grad(_lib.ptr(mp), _lib.ptr(mt), _lib.ptr(mw), _lib.ptr(rows), P, HW, _lib.ptr(coef),
                                         _lib.ptr(grad), _lib.stream_ptr()), "ph_mask_loss_grad")


# ---- one stage -----------------------------------------------------------------------------------------------------------------
def stage_losses(head, cls_score, mask_pred, depth_pred, labels, label_weights, mask_targets, mask_weights, depth_targets,
                 depth_weights, with_grads=False):
    """KernelUpdateHead.loss (kernel_update_head.py:355-441).  cls_score [B, N, L], mask_pred / depth_pred [B, N, H, W] (at the
    assign stride), targets as `get_targets` returns them.  Returns the reference's dict of losses (+ 'pos_acc'); with
    `with_grads` also dict(mask_pred=, cls_score=, depth_pred=) = d(sum of the losses) / d(prediction)."""
    _gpu(mask_pred, "mask_pred")
    lib, dev = _lib.load(), mask_pred.device
    B, N, H, W = mask_pred.shape
    R, HW, L = B * N, H * W, head.num_classes
    mp, dp, cs = _f32(mask_pred).reshape(R, HW), _f32(depth_pred).reshape(R, HW), _f32(cls_score).reshape(R, -1)
    labels = labels.to(dev).long().contiguous()
    pos = (labels >= 0) & (labels < L)                                                        # :375
    num_pos = int(pos.sum())
    from .dist import reduce_mean
    avg = max(float(reduce_mean(pos.sum().float(), _REDUCE_GROUP[-1])), 1.0)                  # :376-377 (mean over ranks, clamp)
    losses, grads = {}, None
    if with_grads:
        grads = dict(mask_pred=torch.empty((R, HW), dtype=torch.float32, device=dev),
                     cls_score=torch.empty_like(cs), depth_pred=torch.empty((R, HW), dtype=torch.float32, device=dev))
    # ---- depth (:383-391)
    ld = head.loss_depth
    mode = DEPTH_MODES[ld.depth_act_mode]
    dt, dw = _f32(depth_targets).reshape(R, HW), _f32(depth_weights).reshape(R, HW)
    w3 = [float(v) for v in ld.weight]
    loss_d, dcoef = _depth_from_sums(depth_loss_sums(dp, dt, dw, mode), ld.loss_weight, w3)
    losses["loss_depth"] = loss_d.float()
    if with_grads:
        _lib.check(lib.ph_depth_loss_grad(_lib.ptr(dp), _lib.ptr(dt), _lib.ptr(dw), R * HW, mode, *dcoef, _lib.ptr(grads["depth_pred"]),
                                          _lib.stream_ptr()), "ph_depth_loss_grad")
    # ---- classification (:393-402)
    lc = head.loss_cls
    lw = _f32(label_weights).reshape(R, -1)
    losses["loss_cls"] = (lc.loss_weight * focal_loss_sum(cs, labels, lw, lc.gamma, lc.alpha) / avg).float()
    if num_pos:
        losses["pos_acc"] = (cs[pos].argmax(1) == labels[pos]).float().sum() * (100.0 / num_pos)       # mmdet accuracy, top-1
    else:
        losses["pos_acc"] = torch.zeros((), device=dev)
    if with_grads:
        _lib.check(lib.ph_focal_loss_grad(_lib.ptr(cs), _lib.ptr(labels), _lib.ptr(lw), R, cs.shape[1], lc.gamma, lc.alpha,
                                          lc.loss_weight / avg, _lib.ptr(grads["cls_score"]), _lib.stream_ptr()), "ph_focal_loss_grad")
    # ---- masks (:404-437)
    _mask_terms(head, mp, mask_targets, mask_weights, pos, B, N, H, W, losses, grads["mask_pred"] if with_grads else None,
                ("loss_rpn_mask", "loss_rpn_dice", "loss_rank"), ("loss_mask", "loss_dice", "loss_rank"))
    if with_grads:
        grads = dict(mask_pred=grads["mask_pred"].reshape(B, N, H, W), cls_score=grads["cls_score"].reshape(B, N, -1),
                     depth_pred=grads["depth_pred"].reshape(B, N, H, W))
        return losses, grads
    return losses


# ---- targets (kernel_update_head.py:443-591) ------------------------------------------------------------------------------------
def target_single(head, pos_inds, neg_inds, pos_mask, neg_mask, pos_gt_mask, pos_gt_labels, gt_sem_seg, gt_sem_cls, pos_depth,
                  neg_depth, gt_depth, gt_valid, cfg):
    """_get_target_single (:443-531): labels / weights / mask and depth targets of ONE image, tensors stay on their device"""
    dev = pos_mask.device
    num_pos, num_neg = pos_mask.shape[0], neg_mask.shape[0]
    R = num_pos + num_neg
    H, W = pos_mask.shape[-2:]
    L, ns, nt = head.num_classes, head.num_stuff_classes, head.num_thing_classes
    pw = 1.0 if cfg.pos_weight <= 0 else cfg.pos_weight
    valid = gt_valid.to(dev).float()
    labels = torch.full((R,), L, dtype=torch.long, device=dev)
    label_weights = torch.zeros((R, L), device=dev)
    mask_targets = torch.zeros((R, H, W), device=dev)
    mask_weights = valid[None].expand(R, H, W).clone()                    # 1 on the valid pixels of every row
    if num_pos:
        labels[pos_inds] = pos_gt_labels
        label_weights[pos_inds] = pw
        mask_targets[pos_inds] = pos_gt_mask.float()
    if num_neg:
        label_weights[neg_inds] = 1.0
    sem_rows = None
    if gt_sem_cls is not None and gt_sem_seg is not None:
        sem_labels = torch.full((ns,), L, dtype=torch.long, device=dev)
        sem_targets = torch.zeros((ns, H, W), device=dev)
        sem_weights = torch.zeros((ns, H, W), device=dev)
        sem_label_weights = torch.cat([torch.zeros((ns, nt), device=dev), torch.eye(ns, device=dev)], -1)
        if len(gt_sem_cls) > 0:
            sem_rows = (gt_sem_cls - nt).long()
            sem_labels[sem_rows] = gt_sem_cls.long()
            sem_targets[sem_rows] = gt_sem_seg.float()
            sem_weights[sem_rows] = 1
        sem_weights = sem_weights * valid
        label_weights[:, nt:] = 0
        labels = torch.cat([labels, sem_labels])
        label_weights = torch.cat([label_weights, sem_label_weights])
        mask_targets = torch.cat([mask_targets, sem_targets])
        mask_weights = torch.cat([mask_weights, sem_weights])
    depth_targets = depth_weights = None
    if pos_depth is not None:
        assert neg_depth is not None and gt_depth is not None
        Rd = R + ns
        gd = gt_depth.to(dev).float().reshape(H, W)           # [H, W] or [1, H, W] (the batched gt_depth of forward_train)
        depth_targets = torch.zeros((Rd, H, W), device=dev)
        depth_weights = torch.zeros((Rd, H, W), device=dev)
        if num_pos:
            depth_targets[pos_inds] = gd
            depth_weights[pos_inds] = pw * pos_gt_mask.float()
        if sem_rows is not None:
            depth_targets[sem_rows + R] = gd
            depth_weights[sem_rows + R] = gt_sem_seg.float() * pw
        depth_targets[-1] = gd                                            # the direct depth row (:525-528)
        depth_weights[-1] = 1.0
        depth_weights = depth_weights * (gd > 0.0).float()
    return labels, label_weights, mask_targets, mask_weights, depth_targets, depth_weights


def get_targets(head, sampling_results, gt_mask, gt_labels, rcnn_train_cfg, concat=True, gt_sem_seg=None, gt_sem_cls=None,
                gt_depth=None):
    """KernelUpdateHead.get_targets (:533-591)"""
    n = len(sampling_results)
    if gt_sem_seg is None:
        gt_sem_seg, gt_sem_cls = [None] * n, [None] * n
    has_depth = gt_depth is not None
    outs = []
    for i, res in enumerate(sampling_results):
        outs.append(target_single(head, res.pos_inds, res.neg_inds, res.pos_masks, res.neg_masks, res.pos_gt_masks, res.pos_gt_labels,
                                  gt_sem_seg[i], gt_sem_cls[i], res.pos_depth if has_depth else None,
                                  res.neg_depth if has_depth else None, gt_depth[i] if has_depth else None, res.valid_mask,
                                  rcnn_train_cfg))
    cols = list(zip(*outs))
    if concat:
        cols = [torch.cat(c, 0) if c[0] is not None else None for c in cols]
    return tuple(cols)



# ---- KernelHead (the rpn side): kernel_head.py:456-698 --------------------------------------------------------------------------
def rpn_losses(head, mask_pred, seg_preds, depth_pred, labels, label_weights, mask_targets, mask_weights, seg_targets,
               depth_targets, depth_weights, with_grads=False):
    """KernelHead.loss (kernel_head.py:456-569) for the shipped configuration (no cls_scores, no semantic_aspp): loss_depth on
    the direct depth map broadcast over the N rows, loss_rpn_mask / loss_rpn_dice / loss_rpn_rank on the positive rows,
    loss_rpn_seg = sigmoid focal loss of conv_seg's map over the labelled pixels.  mask_pred [B, N, H, W], seg_preds
    [B, L, H, W], depth_pred [B, 1 or N + stuff rows, H, W].  With `with_grads`: dict(mask_pred=, seg_preds=, depth_pred= [B, 1, H, W])."""
    _gpu(mask_pred, "mask_pred")
    lib, dev = _lib.load(), mask_pred.device
    B, N, H, W = mask_pred.shape
    R, HW, L = B * N, H * W, head.num_classes
    mp = _f32(mask_pred).reshape(R, HW)
    labels = labels.to(dev).long().contiguous()
    pos = (labels >= 0) & (labels < L)                                                        # :475
    losses, grads = {}, None
    if with_grads:
        grads = dict(mask_pred=torch.empty((R, HW), dtype=torch.float32, device=dev))
    if depth_pred is not None:                                                                # :478-486
        ld = head.loss_depth
        mode = DEPTH_MODES[ld.depth_act_mode]
        Rd = depth_targets.shape[0]               # B * (proposals + stuff rows): one broadcast map per image (:386,482)
        dp = _f32(depth_pred.expand(B, Rd // B, H, W)).reshape(Rd, HW)
        dt, dw = _f32(depth_targets).reshape(Rd, HW), _f32(depth_weights).reshape(Rd, HW)
        loss_d, dcoef = _depth_from_sums(depth_loss_sums(dp, dt, dw, mode), ld.loss_weight, [float(v) for v in ld.weight])
        losses["loss_depth"] = loss_d.float()
        if with_grads:
            g = torch.empty((Rd, HW), dtype=torch.float32, device=dev)
            _lib.check(lib.ph_depth_loss_grad(_lib.ptr(dp), _lib.ptr(dt), _lib.ptr(dw), Rd * HW, mode, *dcoef, _lib.ptr(g),
                                              _lib.stream_ptr()), "ph_depth_loss_grad")
            grads["depth_pred"] = g.reshape(B, Rd // B, H, W).sum(1, keepdim=True)           # the rows are one broadcast map
    _mask_terms(head, mp, mask_targets, mask_weights, pos, B, N, H, W, losses, grads["mask_pred"] if with_grads else None,
                ("loss_rpn_mask", "loss_rpn_dice", "loss_rpn_rank"), ("loss_rpn_mask", "loss_rpn_dice", "loss_rank"))
    if seg_preds is not None:                                                                 # :538-551
        ls = head.loss_seg
        if not ls.use_sigmoid:
            raise NotImplementedError("libpolyhead: loss_seg is the shipped sigmoid FocalLoss (polyphonic_former.py:73-78)")
        sp = _f32(seg_preds).reshape(B, seg_preds.shape[1], HW)
        st = seg_targets.to(dev).reshape(B, HW)
        nd = max(float(((st >= 0) & (st < L)).sum()), 1.0)                                    # num_dense_pos.clamp(min=1)
        st = st.to(torch.int32).contiguous()
        losses["loss_rpn_seg"] = (ls.loss_weight * seg_focal_sum(sp, st, ls.gamma, ls.alpha) / nd).float()
        if with_grads:
            g = torch.empty_like(sp)
            _lib.check(lib.ph_seg_focal_grad(_lib.ptr(sp), _lib.ptr(st), B, sp.shape[1], HW, ls.gamma, ls.alpha, ls.loss_weight / nd,
                                             _lib.ptr(g), _lib.stream_ptr()), "ph_seg_focal_grad")
            grads["seg_preds"] = g.reshape(seg_preds.shape)
    if with_grads:
        grads["mask_pred"] = grads["mask_pred"].reshape(B, N, H, W)
        return losses, grads
    return losses


def dense_depth_loss(head, depth_pred, gt_depth, with_grad=False):
    """losses['depth_dense'] (kernel_head.py:438-442): DepthLoss of the direct depth map against the ground truth, weight = gt > 0"""
    ld = head.loss_depth
    mode = DEPTH_MODES[ld.depth_act_mode]
    dp, gd = _f32(depth_pred).reshape(-1), _f32(gt_depth.to(depth_pred.device)).reshape(-1)
    if dp.numel() != gd.numel():
        raise ValueError(f"depth_dense: prediction {tuple(depth_pred.shape)} and gt_depth {tuple(gt_depth.shape)} differ in size")
    w = (gd > 0).float()
    loss, coef = _depth_from_sums(depth_loss_sums(dp, gd, w, mode), ld.loss_weight, [float(v) for v in ld.weight])
    if not with_grad:
        return loss.float()
    g = torch.empty_like(dp)
    _lib.check(_lib.load().ph_depth_loss_grad(_lib.ptr(dp), _lib.ptr(gd), _lib.ptr(w), dp.numel(), mode, *coef, _lib.ptr(g),
                                              _lib.stream_ptr()), "ph_depth_loss_grad")
    return loss.float(), g.reshape(depth_pred.shape)


def rpn_target_single(head, pos_inds, neg_inds, pos_mask, neg_mask, pos_gt_mask, pos_gt_labels, gt_sem_seg, gt_sem_cls, pos_depth,
                      neg_depth, gt_depth, gt_valid, cfg):
    """KernelHead._get_target_single (kernel_head.py:571-647): ONE image.  Unlike the update heads there are no stuff rows in
    labels / masks (label_weights is per row), a dense `seg_targets` map [H, W] (stuff classes first, then the assigned thing
    masks painted over them in order) and depth rows for the N proposals + the stuff rows, without a direct-depth row."""
    _gpu(pos_mask, "sampling result")
    dev = pos_mask.device
    num_pos, num_neg = pos_mask.shape[0], neg_mask.shape[0]
    R = num_pos + num_neg
    H, W = pos_mask.shape[-2:]
    L, ns, nt = head.num_classes, head.num_stuff_classes, head.num_thing_classes
    pw = 1.0 if cfg.pos_weight <= 0 else cfg.pos_weight
    valid = gt_valid.to(dev).float()
    labels = torch.full((R,), L, dtype=torch.long, device=dev)
    label_weights = torch.zeros((R,), device=dev)
    mask_targets = torch.zeros((R, H, W), device=dev)
    mask_weights = valid[None].expand(R, H, W).clone()
    has_sem = gt_sem_cls is not None and gt_sem_seg is not None
    S = len(gt_sem_cls) if has_sem else 0
    if num_pos:
        labels[pos_inds] = pos_gt_labels
        label_weights[pos_inds] = pw
        mask_targets[pos_inds] = pos_gt_mask.float()
    # the dense semantic target in one pass: stuff classes in order, then the assigned thing masks in order (:590-605)
    seg_targets = torch.empty((H, W), dtype=torch.long, device=dev)
    sem_f = gt_sem_seg.to(dev).float().contiguous() if S else None
    sem_c = gt_sem_cls.to(dev).long().contiguous() if S else None
    pgm = pos_gt_mask.float().contiguous() if num_pos else None
    pgl = pos_gt_labels.to(dev).long().contiguous() if num_pos else None
    _lib.check(_lib.load().ph_seg_target(_lib.ptr(sem_f), _lib.ptr(sem_c), S, _lib.ptr(pgm), _lib.ptr(pgl), num_pos, L, H * W,
                                         _lib.ptr(seg_targets), _lib.stream_ptr()), "ph_seg_target")
    if num_neg:
        label_weights[neg_inds] = 1.0
    depth_targets = depth_weights = None
    if pos_depth is not None:                                                                 # :610-640
        assert neg_depth is not None and gt_depth is not None
        gd = gt_depth.to(dev).float().reshape(H, W)
        depth_targets = torch.zeros((R + ns, H, W), device=dev)
        depth_weights = torch.zeros((R + ns, H, W), device=dev)
        if num_pos:
            depth_targets[pos_inds] = gd
            depth_weights[pos_inds] = pw * pos_gt_mask.float()
        if has_sem and len(gt_sem_cls) > 0:
            rows = (gt_sem_cls - nt).long() + R
            depth_targets[rows] = gd
            depth_weights[rows] = gt_sem_seg.float() * pw
        depth_weights = depth_weights * (gd > 0.0).float()
    return labels, label_weights, mask_targets, mask_weights, seg_targets, depth_targets, depth_weights


def rpn_get_targets(head, sampling_results, gt_mask, rpn_train_cfg, concat=True, gt_sem_seg=None, gt_sem_cls=None, gt_depth=None):
    """KernelHead.get_targets (kernel_head.py:649-698): per image, concatenated (seg_targets stacked)"""
    n = len(sampling_results)
    if gt_sem_seg is None:
        gt_sem_seg, gt_sem_cls = [None] * n, [None] * n
    has_depth = gt_depth is not None
    outs = []
    for i, res in enumerate(sampling_results):
        outs.append(rpn_target_single(head, res.pos_inds, res.neg_inds, res.pos_masks, res.neg_masks, res.pos_gt_masks,
                                      res.pos_gt_labels, gt_sem_seg[i], gt_sem_cls[i], res.pos_depth if has_depth else None,
                                      res.neg_depth if has_depth else None, gt_depth[i] if has_depth else None, res.valid_mask,
                                      rpn_train_cfg))
    cols = list(zip(*outs))
    if concat:
        cols = [None if c[0] is None else (torch.stack(c, 0) if k == 4 else torch.cat(c, 0)) for k, c in enumerate(cols)]
    return tuple(cols)


# =================================================================================================================================
# Round 5: the training path's targets + losses from DESCRIPTORS (csrc/ph_loss.hip `ph_train_losses`; include/polyhead.h).
# `get_targets` / `loss` above stay what the reference's API exposes (materialised [rows][H][W] targets, bit for bit); the
# heads' `forward_train` -- `train.rpn_forward_train` / `train.roi_forward_train` -- no longer goes through them: the Hungarian
# result lives on the host, every target row is a ground-truth row that already lies in HBM, so the host writes pointer tables
# (numpy, ~100 us), uploads them in one copy and ONE C call evaluates the head's or stage's losses and gradients.
# =================================================================================================================================
import ctypes as C  # noqa: E402

import numpy as np  # noqa: E402


class LossCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("B", "N", "L", "P", "depth_rows", "seg_L", "has_rank", "ignore", "depth_mode")] + \
               [("HW", C.c_int64)] + \
               [(n, C.c_float) for n in ("lw_mask", "lw_dice", "dice_eps", "lw_rank", "lw_depth", "dw_si", "dw_sq", "dw_abs", "lw_cls",
                                         "cls_gamma", "cls_alpha", "cls_avg", "lw_seg", "seg_gamma", "seg_alpha")]


class StepGT:
    """one training step's ground truth where the descriptors point: per image contiguous fp32 masks [G, H, W] (hardened when
    `hard_target`), stuff masks [S, H, W], the valid map (any mask set: kernel_head.py:415, kernel_update.py:238), the depth
    map; labels / stuff classes also on the host; the zero-padded stack [B, Gmax, H, W] the batched assignment reads"""

    def __init__(self, gt_masks, gt_labels, gt_sem_seg, gt_sem_cls, gt_depth, hard_target):
        B = len(gt_masks)
        dev = gt_masks[0].device
        f = lambda t: t.detach().float().contiguous()
        self.masks = [f(m.bool()) if hard_target else f(m) for m in gt_masks]
        self.has_sem = gt_sem_seg is not None and gt_sem_cls is not None
        self.sem = [f(s) for s in gt_sem_seg] if self.has_sem else [self.masks[0][:0]] * B
        self.H, self.W = self.masks[0].shape[-2:]
        HW = self.HW = self.H * self.W
        self.G = [int(m.shape[0]) for m in self.masks]
        self.S = [int(s.shape[0]) for s in self.sem]
        valid = torch.zeros((B, self.H, self.W), dtype=torch.float32, device=dev)
        for i in range(B):
            for t in (self.masks[i], self.sem[i]):
                if t.shape[0]:
                    valid[i] += t.ne(0).any(dim=0)
        self.valid = (valid > 0).float()
        if isinstance(gt_depth, (list, tuple)):
            gt_depth = torch.stack([d.reshape(self.H, self.W) for d in gt_depth])
        self.depth = None if gt_depth is None else f(gt_depth.to(dev)).reshape(B, HW)
        # host copies of the small integer lists (ONE synchronising read per step)
        lab = torch.cat([l.reshape(-1).long() for l in gt_labels] + ([c.reshape(-1).long() for c in gt_sem_cls] if self.has_sem else []))
        lab = lab.cpu().numpy()
        o = np.cumsum([0] + self.G + (self.S if self.has_sem else []))
        self.labels_h = [lab[o[i]:o[i + 1]] for i in range(B)]
        self.sem_cls_h = [lab[o[B + i]:o[B + i + 1]] for i in range(B)] if self.has_sem else [np.zeros(0, np.int64)] * B
        self.labels_dev = [l.reshape(-1).long() for l in gt_labels]
        self.Gmax = max(self.G + [0])
        self.pad = torch.zeros((B, max(self.Gmax, 1), self.H, self.W), dtype=torch.float32, device=dev)
        for i in range(B):
            if self.G[i]:
                self.pad[i, :self.G[i]] = self.masks[i]
        self.mask_base = np.array([m.data_ptr() for m in self.masks], dtype=np.int64)
        self.sem_base = np.array([s.data_ptr() for s in self.sem], dtype=np.int64)
        self.valid_base = self.valid.data_ptr() + 4 * HW * np.arange(B, dtype=np.int64)
        self.depth_base = None if self.depth is None else self.depth.data_ptr() + 4 * HW * np.arange(B, dtype=np.int64)
        self.B = B


def assign_batch(assigner, pred, cls_pred, gt):
    """the Hungarian assignment of B images from ONE `ph_match_sums` pass and ONE device -> host copy (funcs/assigner.py:363-542
    per image).  pred [B, Np, H, W] detached mask logits, cls_pred [B, Np, n_thing] or None.  -> per image (pred indices
    ascending, matched gt indices) as numpy arrays.  Configurations the batched algebra does not cover (DepthCost with a weight,
    topk > 1) are refused by the caller."""
    from .assigner import MatchSums, _hungarian
    B, Np = pred.shape[:2]
    out = [(np.zeros(0, np.int64), np.zeros(0, np.int64))] * B
    if gt.Gmax == 0:
        return out
    s = MatchSums(pred, gt.pad, gt.valid)
    cost = 0
    a = assigner
    if a.cls_cost.weight != 0 and cls_pred is not None:
        cc = a.cls_cost
        p = cls_pred.sigmoid()
        neg = -(1 - p + cc.eps).log() * (1 - cc.alpha) * p.pow(cc.gamma)
        pos = -(p + cc.eps).log() * cc.alpha * (1 - p).pow(cc.gamma)
        lab = torch.zeros((B, gt.Gmax), dtype=torch.long, device=pred.device)
        for i in range(B):
            if gt.G[i]:
                lab[i, :gt.G[i]] = gt.labels_dev[i]
        idx = lab[:, None, :].expand(B, Np, gt.Gmax)
        cost = cost + (pos.gather(2, idx) - neg.gather(2, idx)) * cc.weight
    if a.mask_cost.weight != 0:
        V = s.V[:, None, None]
        cost = cost + (-(s.A + (V - s.S[:, :, None] - s.T[:, None, :] + s.A)) / V * a.mask_cost.weight)
    if a.dice_cost.weight != 0:
        e = a.dice_cost.eps
        cost = cost + (-(2 * s.A) / ((s.Q + e)[:, :, None] + (s.C + e)[:, None, :]) * a.dice_cost.weight)
    cost = cost.detach().cpu().numpy()
    for i in range(B):
        if gt.G[i]:
            r, c = _hungarian(torch.from_numpy(cost[i, :, :gt.G[i]]), 1)
            order = np.argsort(r, kind="stable")
            out[i] = (np.asarray(r)[order].astype(np.int64), np.asarray(c)[order].astype(np.int64))
    return out


class LossDesc:
    """the pointer tables of one head / stage on the device (one upload) + what the host knows about them"""

    def __init__(self, dev, sections):
        off, total = {}, 0
        for k, a in sections.items():
            off[k] = total
            total += (a.nbytes + 15) // 16 * 16
        buf = np.zeros(max(total, 16), dtype=np.uint8)
        for k, a in sections.items():
            buf[off[k]:off[k] + a.nbytes] = a.view(np.uint8).reshape(-1)
        self.blob = torch.from_numpy(buf).to(dev, non_blocking=True)
        base = self.blob.data_ptr()
        self.ptr = {k: C.c_void_p(base + o) for k, o in off.items()}
        self.n = {k: a.size for k, a in sections.items()}


def build_desc(head, gt, assigns, num_proposals, cfg, roi):
    """roi=True : KernelUpdateHead._get_target_single (kernel_update_head.py:443-531) for B images: N = num_proposals + stuff rows
    roi=False: KernelHead._get_target_single (kernel_head.py:571-647): N = num_proposals rows, dense semantic target, the depth
    items of an image all on its ONE direct depth map"""
    B, HW = gt.B, gt.HW
    L, ns, nt = head.num_classes, head.num_stuff_classes, head.num_thing_classes
    Np = num_proposals
    N = Np + ns if (roi and gt.has_sem) else Np
    R = B * N
    pw = 1.0 if cfg.pos_weight <= 0 else float(cfg.pos_weight)
    HW4 = 4 * HW
    tptr, wptr = np.zeros(R, np.int64), np.zeros(R, np.int64)
    labels = np.full(R, L, np.int64)
    pos_u8 = np.zeros(R, np.uint8)
    label_w = np.zeros((R, L), np.float32) if roi else None
    d_row, d_t, d_w, d_s = [], [], [], []
    s_cnt, s_m, s_l = [0], [], []
    for b in range(B):
        r0 = b * N
        pi, gi = assigns[b]
        rows = r0 + pi
        wptr[r0:r0 + Np] = gt.valid_base[b]
        tptr[rows] = gt.mask_base[b] + gi * HW4
        labels[rows] = gt.labels_h[b][gi]
        pos_u8[rows] = 1
        if roi:
            label_w[r0:r0 + Np, :(nt if gt.has_sem else L)] = 1.0
            label_w[rows, :(nt if gt.has_sem else L)] = pw
        sc = gt.sem_cls_h[b]
        srow = None
        if roi and gt.has_sem:
            label_w[r0 + Np + np.arange(ns), nt + np.arange(ns)] = 1.0
            if len(sc):
                srow = r0 + Np + (sc - nt)
                tptr[srow] = gt.sem_base[b] + np.arange(len(sc), dtype=np.int64) * HW4
                wptr[srow] = gt.valid_base[b]
                labels[srow] = sc
                pos_u8[srow] = 1
        if gt.depth_base is not None:
            dp = int(gt.depth_base[b])
            if roi:
                last = r0 + N - 1
                for r, w in zip(rows.tolist(), (gt.mask_base[b] + gi * HW4).tolist()):
                    if r != last:
                        d_row.append(r); d_t.append(dp); d_w.append(w); d_s.append(pw)
                if srow is not None:
                    for k, r in enumerate(srow.tolist()):
                        if r != last:
                            d_row.append(r); d_t.append(dp); d_w.append(int(gt.sem_base[b]) + k * HW4); d_s.append(pw)
                d_row.append(last); d_t.append(dp); d_w.append(1); d_s.append(1.0)          # the direct depth row (:525-528)
            else:
                for w in (gt.mask_base[b] + gi * HW4).tolist():
                    d_row.append(b); d_t.append(dp); d_w.append(w); d_s.append(pw)
                if gt.has_sem:
                    for k in range(len(sc)):
                        d_row.append(b); d_t.append(dp); d_w.append(int(gt.sem_base[b]) + k * HW4); d_s.append(pw)
        if not roi:         # dense semantic target: the stuff masks in order, then the assigned thing masks in order (:590-605)
            if gt.has_sem:
                for k in range(len(sc)):
                    s_m.append(int(gt.sem_base[b]) + k * HW4); s_l.append(int(sc[k]))
            for w, l in zip((gt.mask_base[b] + gi * HW4).tolist(), gt.labels_h[b][gi].tolist()):
                s_m.append(w); s_l.append(int(l))
            s_cnt.append(len(s_m))
    depth_rows = R if roi else B
    d_row = np.asarray(d_row, np.int64)
    order = np.argsort(d_row, kind="stable")
    dstart = np.zeros(depth_rows + 1, np.int32)
    if len(d_row):
        np.add.at(dstart, d_row + 1, 1)
    dstart = np.cumsum(dstart).astype(np.int32)
    pos_rows = np.nonzero(pos_u8)[0].astype(np.int32)
    sec = dict(tptr=tptr, wptr=wptr, labels=labels, pos_u8=pos_u8, pos_rows=pos_rows if len(pos_rows) else np.zeros(1, np.int32), dstart=dstart,
               dit_t=np.asarray(d_t, np.int64)[order] if len(d_row) else np.zeros(1, np.int64),
               dit_w=np.asarray(d_w, np.int64)[order] if len(d_row) else np.zeros(1, np.int64),
               dit_s=np.asarray(d_s, np.float32)[order] if len(d_row) else np.zeros(1, np.float32))
    if roi:
        sec["label_w"] = label_w.reshape(-1)
    else:
        sec.update(sstart=np.asarray(s_cnt, np.int32), sit_m=np.asarray(s_m if s_m else [0], np.int64), sit_l=np.asarray(s_l if s_l else [0], np.int32))
    d = LossDesc(gt.valid.device, sec)
    d.B, d.N, d.R, d.P, d.depth_rows, d.roi, d.has_depth = B, N, R, int(len(pos_rows)), depth_rows, roi, gt.depth_base is not None
    d.gt = gt              # keeps the ground-truth tensors the pointers address alive
    return d


_LOSS_SCRATCH = {}


def fused_losses(head, desc, mask_pred, cls_score, depth_pred, seg_pred, with_grads=True):
    """`ph_train_losses` for one head / stage: -> (dict of weighted losses as the reference names them, grads or None).
    mask_pred [B, N, H, W]; cls_score [B, N, L] (roi) / None; depth_pred [B, N, H, W] (roi) / [B, 1, H, W] (KernelHead's direct
    map) / None; seg_pred [B, L, H, W] (KernelHead) / None"""
    lib, dev = _lib.load(), mask_pred.device
    B, N, H, W = mask_pred.shape
    assert (B, N) == (desc.B, desc.N)
    HW, L = H * W, head.num_classes
    roi = desc.roi
    mp = _f32(mask_pred)
    cs = _f32(cls_score).reshape(B * N, L) if cls_score is not None else None
    have_depth = depth_pred is not None and desc.has_depth
    dp = _f32(depth_pred) if have_depth else mp                 # a valid pointer either way; without items nothing is read
    sp = _f32(seg_pred) if seg_pred is not None else None
    c = LossCfg()
    c.B, c.N, c.L, c.P, c.depth_rows, c.HW = B, N, L, desc.P, desc.depth_rows, HW
    c.seg_L = sp.shape[1] if sp is not None else 0
    lr = head.loss_rank
    c.has_rank, c.ignore = int(lr is not None), int(head.ignore_label)
    c.lw_mask, c.lw_dice, c.dice_eps = head.loss_mask.loss_weight, head.loss_dice.loss_weight, head.loss_dice.eps
    c.lw_rank = lr.loss_weight if lr is not None else 0.0
    ld = head.loss_depth
    if have_depth:
        c.depth_mode = DEPTH_MODES[ld.depth_act_mode]
        c.lw_depth = ld.loss_weight
        c.dw_si, c.dw_sq, c.dw_abs = [float(v) for v in ld.weight]
    if cs is not None:
        lc = head.loss_cls
        from .dist import reduce_mean
        import torch.distributed as tdist
        npos = float(desc.P)
        if tdist.is_available() and tdist.is_initialized() and tdist.get_world_size(_REDUCE_GROUP[-1]) > 1:
            # gloo groups (the CPU tests) reduce host tensors, RCCL groups device tensors
            on = dev if tdist.get_backend(_REDUCE_GROUP[-1]) == "nccl" else "cpu"
            npos = float(reduce_mean(torch.tensor(npos, device=on), _REDUCE_GROUP[-1]))
        c.lw_cls, c.cls_gamma, c.cls_alpha, c.cls_avg = lc.loss_weight, lc.gamma, lc.alpha, max(npos, 1.0)
    else:
        c.cls_avg = 1.0
    if sp is not None:
        ls = head.loss_seg
        if not ls.use_sigmoid:
            raise NotImplementedError("libpolyhead: loss_seg is the shipped sigmoid FocalLoss (polyphonic_former.py:73-78)")
        c.lw_seg, c.seg_gamma, c.seg_alpha = ls.loss_weight, ls.gamma, ls.alpha
    nb = lib.ph_train_losses_scratch_bytes(C.byref(c))
    scratch = _LOSS_SCRATCH.get(dev)
    if scratch is None or scratch.numel() < nb:
        scratch = _LOSS_SCRATCH[dev] = torch.empty((nb,), dtype=torch.uint8, device=dev)
    out = torch.empty((8,), dtype=torch.float32, device=dev)
    g = None
    if with_grads:
        g = dict(mask_pred=torch.empty_like(mp), depth_pred=torch.empty_like(dp) if have_depth else None,
                 cls_score=torch.empty_like(cs) if cs is not None else None, seg_preds=torch.empty_like(sp) if sp is not None else None)
    P = desc.ptr
    gd = g["depth_pred"] if (g and have_depth) else (torch.empty_like(dp) if g else None)
    if not have_depth:        # no depth items: an empty item table (dstart all zero) -- the kernel reads nothing
        pass
    _lib.check(lib.ph_train_losses(C.byref(c), _lib.ptr(mp), _lib.ptr(cs), _lib.ptr(dp), _lib.ptr(sp), P["pos_rows"], P["pos_u8"], P["tptr"],
                                   P["wptr"], P["dstart"], P["dit_t"], P["dit_w"], P["dit_s"], P["labels"] if cs is not None else None,
                                   P.get("label_w") if cs is not None else None, P.get("sstart") if sp is not None else None,
                                   P.get("sit_m") if sp is not None else None, P.get("sit_l") if sp is not None else None, _lib.ptr(out),
                                   _lib.ptr(g["mask_pred"]) if g else None, _lib.ptr(g["cls_score"]) if (g and cs is not None) else None,
                                   _lib.ptr(gd), _lib.ptr(g["seg_preds"]) if (g and sp is not None) else None, _lib.ptr(scratch),
                                   scratch.numel(), _lib.stream_ptr()), "ph_train_losses")
    losses = {}
    if have_depth:
        losses["loss_depth"] = out[0]
    if roi:
        losses["loss_cls"], losses["pos_acc"] = out[1], out[6]
        keys = ("loss_rpn_mask", "loss_rpn_dice", "loss_rank") if desc.P else ("loss_mask", "loss_dice", "loss_rank")
    else:
        keys = ("loss_rpn_mask", "loss_rpn_dice", "loss_rpn_rank") if desc.P else ("loss_rpn_mask", "loss_rpn_dice", "loss_rank")
    losses[keys[0]], losses[keys[1]] = out[2], out[3]
    if lr is not None:
        losses[keys[2]] = out[4]
    if sp is not None:
        losses["loss_rpn_seg"] = out[5]
    if g is not None:
        g = dict(mask_pred=g["mask_pred"].reshape(B, N, H, W), cls_score=None if cs is None else g["cls_score"].reshape(B, N, L),
                 depth_pred=None if not have_depth else g["depth_pred"], seg_preds=g["seg_preds"])
    return losses, g
