"""TEST INFRASTRUCTURE ONLY -- CPU restatement (the *oracle*) of the training-side targets and losses of one
KernelUpdateHead stage and of KernelHead (the rpn side).  Plain fp32 torch on the CPU, functional; every function cites the reference lines it follows.
Pinned by tests/golden/loss.npz, which `oracle/gen_golden_loss.py` produced with the reference's own
`KernelUpdateHead.get_targets` / `.loss`, the vendored mmdet FocalLoss / CrossEntropyLoss / DiceLoss / accuracy and the
project's DepthLoss and by tests/golden/train_rpn.npz (the reference's KernelHead.forward_train) (tests/test_loss_oracle.py).  Only tests may
import this file."""
import torch
import torch.nn.functional as F

from .poly_oracle import depth_act


# ---- mmdet/models/losses ---------------------------------------------------------------------------------------------
def focal_loss(pred, labels, weight, avg_factor, gamma=2.0, alpha=0.25, loss_weight=2.0):
    """FocalLoss.forward on the CPU path (focal_loss.py:221-240): one-hot of `labels` over L + 1 classes without the
    background column, py_sigmoid_focal_loss (:12-60), sum / avg_factor (utils.py:29-55)."""
    L = pred.shape[1]
    t = F.one_hot(labels, num_classes=L + 1)[:, :L].type_as(pred)
    p = pred.sigmoid()
    pt = (1 - p) * t + p * (1 - t)
    fw = (alpha * t + (1 - alpha) * (1 - t)) * pt.pow(gamma)
    loss = F.binary_cross_entropy_with_logits(pred, t, reduction="none") * fw
    return loss_weight * (loss * weight).sum() / avg_factor


def bce_mean(pred, target, loss_weight=1.0):
    """CrossEntropyLoss(use_sigmoid=True) on equal-rank inputs (cross_entropy_loss.py:74-113): mean BCE with logits"""
    return loss_weight * F.binary_cross_entropy_with_logits(pred, target.float(), reduction="none").mean()


def dice_one(pred_logits, target, eps=1e-3, loss_weight=4.0):
    """DiceLoss.forward on ONE sample of selected pixels (dice_loss.py:9-46,118-136): sigmoid, 1 - 2a / (b + c)"""
    x = pred_logits.sigmoid().flatten()
    t = target.flatten().float()
    a = (x * t).sum()
    b = (x * x).sum() + eps
    c = (t * t).sum() + eps
    return loss_weight * (1 - 2 * a / (b + c))


def rank_loss(mask_pred, rank_target, ignore=255, loss_weight=0.1):
    """CrossEntropyLoss(use_sigmoid=False): F.cross_entropy(reduction='none', ignore_index) then the mean over ALL pixels,
    ignored ones included as zeros (cross_entropy_loss.py:9-47; this mmdet has no avg_non_ignore)"""
    return loss_weight * F.cross_entropy(mask_pred, rank_target, reduction="none", ignore_index=ignore).mean()


def depth_loss(pred_logits, target, mask_weight, mode="sigmoid", loss_weight=5.0, weights=(1.0, 1.0, 1.0),
               min_depth=0.0, max_depth=80.0):
    """DepthLoss.forward + depth_loss (polyphonic/losses/depth_loss.py:9-65): depth_act, the three error terms over the
    pixels with 0 < target < 80 and non-zero weight, (terms * weight).mean() * loss_weight.  NB the scale-invariant term
    subtracts sum(log_minus) / n^2 as the reference writes it (:25), not the squared sum."""
    pred = depth_act(pred_logits, mode)
    mask = (target > min_depth) & (target < max_depth) & (mask_weight != 0)
    if not torch.any(mask):
        return loss_weight * (torch.zeros(3) * torch.tensor(weights)).mean()
    p, t, w = pred[mask], target[mask], mask_weight[mask]
    n = p.shape[0]
    lm = (torch.log(p) - torch.log(t)) * w
    m = (p - t) * w
    si = (lm ** 2).sum() / n - lm.sum() / (n ** 2)
    sq = torch.sqrt(((m / t) ** 2).sum() / n)
    ab = (m / t).abs().sum() / n
    return loss_weight * (torch.stack((si, sq, ab)) * torch.tensor(weights)).mean()


def accuracy_top1(pred, target):
    """mmdet accuracy(topk=1) in percent (accuracy.py:6-50); 0 for no samples"""
    if pred.shape[0] == 0:
        return pred.new_tensor(0.0)
    return (pred.argmax(1) == target).float().sum() * (100.0 / pred.shape[0])


# ---- KernelUpdateHead._get_target_single / get_targets (kernel_update_head.py:443-591) ---------------------------------
def target_single(num_classes, n_thing, n_stuff, pos_inds, neg_inds, num_samples, H, W, pos_gt_mask, pos_gt_labels, gt_sem_seg,
                  gt_sem_cls, gt_depth, gt_valid, pos_weight=1.0):
    labels = torch.full((num_samples,), num_classes, dtype=torch.long)                       # :459-461
    label_weights = torch.zeros((num_samples, num_classes))
    mask_targets = torch.zeros((num_samples, H, W))
    mask_weights = torch.zeros((num_samples, H, W))
    mask_weights[..., gt_valid.bool()] = 1.0                                                 # :465
    pw = 1.0 if pos_weight <= 0 else pos_weight
    if len(pos_inds):                                                                        # :467-472
        labels[pos_inds] = pos_gt_labels
        label_weights[pos_inds] = pw
        mask_targets[pos_inds] = pos_gt_mask
    if len(neg_inds):
        label_weights[neg_inds] = 1.0                                                        # :474-475
    sem_inds = None
    if gt_sem_cls is not None and gt_sem_seg is not None:                                    # :477-500
        sem_labels = torch.full((n_stuff,), num_classes, dtype=torch.long)
        sem_targets = torch.zeros((n_stuff, H, W))
        sem_weights = torch.zeros((n_stuff, H, W))
        sem_label_weights = torch.cat([torch.zeros((n_stuff, n_thing)), torch.eye(n_stuff)], -1)
        if len(gt_sem_cls) > 0:
            sem_inds = (gt_sem_cls - n_thing).long()
            sem_labels[sem_inds] = gt_sem_cls.long()
            sem_targets[sem_inds] = gt_sem_seg
            sem_weights[sem_inds] = 1
        sem_weights = sem_weights * gt_valid
        label_weights[:, n_thing:] = 0
        labels = torch.cat([labels, sem_labels])
        label_weights = torch.cat([label_weights, sem_label_weights])
        mask_targets = torch.cat([mask_targets, sem_targets])
        mask_weights = torch.cat([mask_weights, sem_weights])
    depth_targets = depth_weights = None
    if gt_depth is not None:                                                                 # :502-531
        R = num_samples + n_stuff
        depth_targets = torch.zeros((R, H, W))
        depth_weights = torch.zeros((R, H, W))
        depth_valid = (gt_depth[None].repeat(R, 1, 1) > 0.0).float()
        if len(pos_inds):
            depth_targets[pos_inds] = gt_depth[None].repeat(len(pos_inds), 1, 1)
            depth_weights[pos_inds] = pw * pos_gt_mask
        if sem_inds is not None:
            depth_targets[sem_inds + num_samples] = gt_depth
            depth_weights[sem_inds + num_samples] = gt_sem_seg * pw
        depth_targets[-1] = gt_depth                                                         # :525-528 (direct depth)
        depth_weights[-1] = 1.0
        depth_weights = depth_weights * depth_valid
    return labels, label_weights, mask_targets, mask_weights, depth_targets, depth_weights


def get_targets(num_classes, n_thing, n_stuff, Nq, H, W, gts, valids):
    """gts: per image dict(masks, labels, sem_seg, sem_cls, depth, gt_inds, assigned_labels) -- the sampling result of
    MaskPseudoSampler is pos = gt_inds > 0, neg = gt_inds == 0 (funcs/sampler.py:81-113).  Concatenated over images."""
    outs = []
    for g, v in zip(gts, valids):
        pos = (g["gt_inds"] > 0).nonzero().flatten()
        neg = (g["gt_inds"] == 0).nonzero().flatten()
        pos_gt = g["masks"][g["gt_inds"][pos] - 1] if len(g["masks"]) else g["masks"][:0]
        outs.append(target_single(num_classes, n_thing, n_stuff, pos, neg, Nq, H, W, pos_gt, g["assigned_labels"][pos], g["sem_seg"],
                                  g["sem_cls"], g["depth"], v))
    return tuple(torch.cat([o[k] for o in outs], 0) for k in range(6))


# ---- KernelUpdateHead.loss (kernel_update_head.py:355-441) ---------------------------------------------------------------
def stage_loss(num_classes, cls_score, mask_pred, depth_pred, labels, label_weights, mask_targets, mask_weights, depth_targets,
               depth_weights, ignore_label=255):
    losses = {}
    pos = (labels >= 0) & (labels < num_classes)                                             # :375
    avg = pos.sum().float().clamp(min=1.0)                                                   # :376-377
    B, N, H, W = mask_pred.shape
    R = B * N
    losses["loss_depth"] = depth_loss(depth_pred.reshape(R, H, W), depth_targets, depth_weights)        # :383-391
    cs = cls_score.reshape(R, -1)
    losses["loss_cls"] = focal_loss(cs, labels, label_weights, avg)                          # :395-400
    losses["pos_acc"] = accuracy_top1(cs[pos], labels[pos])                                  # :401-402
    if pos.any():                                                                            # :408-437
        pm = mask_pred.reshape(R, H, W)[pos]
        pt = mask_targets[pos]
        pw = mask_weights[pos].bool()
        losses["loss_rpn_mask"] = bce_mean(pm[pw], pt[pw])
        losses["loss_rpn_dice"] = torch.stack([dice_one(pm[i][pw[i]], pt[i][pw[i]]) for i in range(pm.shape[0])]).mean()
        rank_target = torch.full((B, H, W), ignore_label, dtype=torch.long)
        mt = mask_targets.view(B, -1, H, W).bool()
        for b, j in pos.view(B, -1).nonzero(as_tuple=False).tolist():
            rank_target[b][mt[b][j]] = j
        losses["loss_rank"] = rank_loss(mask_pred, rank_target, ignore_label)
    else:
        losses["loss_mask"] = mask_pred.sum() * 0
        losses["loss_dice"] = mask_pred.sum() * 0
        losses["loss_rank"] = mask_pred.sum() * 0
    return losses


# ---- KernelHead._get_target_single / get_targets / loss (kernel_head.py:456-698) --------------------------------------------
def rpn_target_single(num_classes, n_thing, n_stuff, pos_inds, neg_inds, num_samples, H, W, pos_gt_mask, pos_gt_labels, gt_sem_seg,
                      gt_sem_cls, gt_depth, gt_valid, pos_weight=1.0):
    labels = torch.full((num_samples,), num_classes, dtype=torch.long)                       # :583-586
    label_weights = torch.zeros(num_samples)
    mask_targets = torch.zeros((num_samples, H, W))
    mask_weights = torch.zeros((num_samples, H, W))
    mask_weights[..., gt_valid.bool()] = 1.0                                                 # :589
    seg_targets = torch.full((H, W), num_classes, dtype=torch.long)
    for sem_mask, sem_cls in zip(gt_sem_seg.bool(), gt_sem_cls):                             # :594-597
        seg_targets[sem_mask] = sem_cls.long()
    pw = 1.0 if pos_weight <= 0 else pos_weight
    if len(pos_inds):                                                                        # :599-605
        labels[pos_inds] = pos_gt_labels
        label_weights[pos_inds] = pw
        mask_targets[pos_inds] = pos_gt_mask
        for i in range(len(pos_inds)):
            seg_targets[pos_gt_mask[i].bool()] = pos_gt_labels[i]
    if len(neg_inds):
        label_weights[neg_inds] = 1.0
    R = num_samples + n_stuff                                                                # :610-640
    depth_targets = torch.zeros((R, H, W))
    depth_weights = torch.zeros((R, H, W))
    depth_valid = (gt_depth.reshape(1, H, W).repeat(R, 1, 1) > 0.0).float()
    if len(pos_inds):
        depth_targets[pos_inds] = gt_depth.reshape(1, H, W).repeat(len(pos_inds), 1, 1)
        depth_weights[pos_inds] = pw * pos_gt_mask
    if len(gt_sem_cls) > 0:
        rows = (gt_sem_cls - n_thing).long() + num_samples
        depth_targets[rows] = gt_depth.reshape(H, W)
        depth_weights[rows] = gt_sem_seg * pw
    depth_weights = depth_weights * depth_valid
    return labels, label_weights, mask_targets, mask_weights, seg_targets, depth_targets, depth_weights


def rpn_get_targets(num_classes, n_thing, n_stuff, Nq, H, W, gts, valids):
    """as `get_targets` above; seg_targets are stacked (kernel_head.py:688)"""
    outs = []
    for g, v in zip(gts, valids):
        pos = (g["gt_inds"] > 0).nonzero().flatten()
        neg = (g["gt_inds"] == 0).nonzero().flatten()
        pos_gt = g["masks"][g["gt_inds"][pos] - 1] if len(g["masks"]) else g["masks"][:0]
        outs.append(rpn_target_single(num_classes, n_thing, n_stuff, pos, neg, Nq, H, W, pos_gt, g["assigned_labels"][pos], g["sem_seg"],
                                      g["sem_cls"], g["depth"], v))
    return tuple(torch.stack([o[k] for o in outs], 0) if k == 4 else torch.cat([o[k] for o in outs], 0) for k in range(7))


def rpn_loss(num_classes, mask_pred, seg_preds, depth_pred, labels, label_weights, mask_targets, mask_weights, seg_targets,
             depth_targets, depth_weights, ignore_label=255):
    """KernelHead.loss with the shipped losses (polyphonic_former.py:66-97): cls_scores and semantic_aspp are None"""
    losses = {}
    pos = (labels >= 0) & (labels < num_classes)                                             # :475
    B, N, H, W = mask_pred.shape
    R = B * N
    Rd = depth_targets.shape[0]                      # B * (proposals + stuff rows): forward_train expands the one map (:386)
    losses["loss_depth"] = depth_loss(depth_pred.expand(B, Rd // B, H, W).reshape(Rd, H, W), depth_targets, depth_weights)   # :478-486
    if pos.any():                                                                            # :503-531
        pm = mask_pred.reshape(R, H, W)[pos]
        pt = mask_targets[pos]
        pw = mask_weights[pos].bool()
        losses["loss_rpn_mask"] = bce_mean(pm[pw], pt[pw])
        losses["loss_rpn_dice"] = torch.stack([dice_one(pm[i][pw[i]], pt[i][pw[i]]) for i in range(pm.shape[0])]).mean()
        rank_target = torch.full((B, H, W), ignore_label, dtype=torch.long)
        mt = mask_targets.view(B, -1, H, W).bool()
        for b, j in pos.view(B, -1).nonzero(as_tuple=False).tolist():
            rank_target[b][mt[b][j]] = j
        losses["loss_rpn_rank"] = rank_loss(mask_pred, rank_target, ignore_label)
    else:                                                                                    # :533-537
        losses["loss_rpn_mask"] = mask_pred.sum() * 0
        losses["loss_rpn_dice"] = mask_pred.sum() * 0
        losses["loss_rank"] = mask_pred.sum() * 0
    L = seg_preds.shape[1]                                                                   # :539-551
    sel = seg_targets != L
    flat = seg_preds.permute(1, 0, 2, 3)[..., sel].permute(1, 0)
    ft = seg_targets[sel]
    nd = ((ft >= 0) & (ft < num_classes)).sum().float().clamp(min=1.0)
    losses["loss_rpn_seg"] = focal_loss(flat, ft, torch.ones(()), nd, loss_weight=1.0)
    return losses


def dense_depth(depth_pred, gt_depth):
    """losses['depth_dense'] (kernel_head.py:438-442)"""
    return depth_loss(depth_pred, gt_depth, (gt_depth > 0).float())
