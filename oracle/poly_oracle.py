"""TEST INFRASTRUCTURE ONLY -- CPU restatement (the *oracle*) of PolyphonicFormer's unified-query
decode hot path.  Plain fp32 PyTorch-CPU / numpy arithmetic, functional, no mmcv, no modules.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
file, and only as the checker / the timed CPU baseline -- never as the product path.  The product
(``polyphonicformer_amd``) fails loudly when its HIP library is missing and has no CPU fallback.

Pinning: the reference ships no tests, fixtures or golden vectors of its own,
so this restatement is pinned against *outputs of the reference itself run in the build container*
(`oracle/gen_golden.py` imports /root/reference through `oracle/ref_loader.py` and writes
``tests/golden/*.npz``); ``tests/test_oracle_golden.py`` replays those vectors through this file.

Every function cites the reference lines it follows.  Weights are passed as a flat dict with the
reference's ``state_dict`` key names, e.g.
``mask_head.0.kernel_update_conv.dynamic_layer.weight``.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

LN_EPS = 1e-5
GN_EPS = 1e-5


# --------------------------------------------------------------------------------------------
# small bricks (mmcv 1.3.18 semantics)
# --------------------------------------------------------------------------------------------
def _lin(sd, name, x, bias=True):
    return F.linear(x, sd[name + ".weight"], sd[name + ".bias"] if bias else None)


def _ln(sd, name, x):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], LN_EPS)


def binarize(logits, thr=0.5):
    """kernel_update_head.py:236-238 / kernel_head.py:314-317: sigmoid -> > thr -> float."""
    return (logits.sigmoid() > thr).float()


def depth_act(z, mode="sigmoid", min_depth=0.01, max_depth=80.0):
    """polyphonic/funcs/depth_utils.py:1-19."""
    if mode == "sigmoid":
        return z.sigmoid() * (max_depth - min_depth) + min_depth
    if mode == "monodepth":
        disp = z.sigmoid()
        return 1.0 / (1.0 / max_depth + (1.0 / min_depth - 1.0 / max_depth) * disp)
    raise NotImplementedError(mode)


def mha_self(sd, name, t, heads):
    """mmcv MultiheadAttention wrapper around nn.MultiheadAttention, q=k=v=t, *including* the
    wrapper's identity (kernel_update_head.py:112-115,259).  t: [B, N, C] (batch-first here; the
    reference runs sequence-first [N, B, C], which is the same arithmetic per image)."""
    B, N, C = t.shape
    d = C // heads
    qkv = F.linear(t, sd[name + ".attn.in_proj_weight"], sd[name + ".attn.in_proj_bias"])
    q, k, v = qkv.split(C, dim=-1)
    q = q.view(B, N, heads, d).transpose(1, 2) * (1.0 / math.sqrt(d))  # torch scales q first
    k = k.view(B, N, heads, d).transpose(1, 2)
    v = v.view(B, N, heads, d).transpose(1, 2)
    p = torch.softmax(q @ k.transpose(-1, -2), dim=-1)
    a = (p @ v).transpose(1, 2).reshape(B, N, C)
    out = F.linear(a, sd[name + ".attn.out_proj.weight"], sd[name + ".attn.out_proj.bias"])
    return t + out


def ffn(sd, name, t):
    """mmcv FFN(num_fcs=2, ReLU, dropout 0, add_identity) -- kernel_update_head.py:146-158,271."""
    h = F.relu(_lin(sd, name + ".layers.0.0", t))
    return t + _lin(sd, name + ".layers.1", h)


# --------------------------------------------------------------------------------------------
# A.3  KernelUpdator.forward  (polyphonic/funcs/kernel_updator.py:55-93)
# --------------------------------------------------------------------------------------------
def kernel_updator(sd, name, u, k):
    """u = pooled feature [.., C], k = kernel [.., C]; conv_kernel_size = 1 so K*K == 1."""
    Cf = sd[name + ".input_gate.weight"].shape[0]
    p = _lin(sd, name + ".dynamic_layer", u)              # :58
    p_in, p_out = p[..., :Cf], p[..., -Cf:]               # :59-62
    i = _lin(sd, name + ".input_layer", k)                # :64-65
    i_in, i_out = i[..., :Cf], i[..., -Cf:]               # :66-67
    g = i_in * p_in                                       # :69
    ig = _ln(sd, name + ".input_norm_in", _lin(sd, name + ".input_gate", g)).sigmoid()   # :73,76
    ug = _ln(sd, name + ".norm_in", _lin(sd, name + ".update_gate", g)).sigmoid()        # :74,77
    p_out = _ln(sd, name + ".norm_out", p_out)            # :78
    i_out = _ln(sd, name + ".input_norm_out", i_out)      # :79
    f = ug * p_out + ig * i_out                           # :86-87
    f = _lin(sd, name + ".fc_layer", f)                   # :89
    return F.relu(_ln(sd, name + ".fc_norm", f))          # :90-91


# --------------------------------------------------------------------------------------------
# A.2  one KernelUpdateHead stage  (polyphonic/kernel_update_head.py:212-353)
# --------------------------------------------------------------------------------------------
def update_stage(sd, pre, x, k, m, q, dfe, heads=8, hard_mask_thr=0.5, hard_mask=None):
    """x, dfe [B,C,H,W]; k, q [B,N,C]; m [B,N,H,W] mask logits (same H,W).
    Returns dict(cls [B,N,L], mask [B,N,H,W], obj [B,N,C], depth [B,N,H,W], dobj [B,N,C]).
    `hard_mask` (tests only): a {0,1} tensor [B,N,H,W] used INSTEAD of binarize(m) -- the hard decisions of another
    implementation's run, so that a free-running comparison measures arithmetic and not the discontinuity."""
    B, C, H, W = x.shape
    N = k.shape[1]
    xt = F.conv2d(x, sd[pre + "feat_transform.conv.weight"], sd[pre + "feat_transform.conv.bias"])          # :225
    dt = F.conv2d(dfe, sd[pre + "feat_depth_transform.conv.weight"], sd[pre + "feat_depth_transform.conv.bias"])  # :226
    M = binarize(m, hard_mask_thr) if hard_mask is None else hard_mask.to(m.dtype)   # :236-238
    u = torch.einsum("bnhw,bchw->bnc", M, xt)                        # :241
    ud = torch.einsum("bnhw,bchw->bnc", M, dt)                       # :242
    q = q + k.detach()                                               # :250 (`.detach()`: no gradient into the mask kernel; same values)
    o = kernel_updator(sd, pre + "kernel_update_conv", u, k)         # :252
    od = kernel_updator(sd, pre + "kernel_update_conv_depth", ud, q)  # :253
    o = _ln(sd, pre + "attention_norm", mha_self(sd, pre + "attention", o, heads))                # :259
    od = _ln(sd, pre + "attention_norm_depth", mha_self(sd, pre + "attention_depth", od, heads))  # :260
    o = _ln(sd, pre + "ffn_norm", ffn(sd, pre + "ffn", o))                                        # :271
    od = _ln(sd, pre + "ffn_norm_depth", ffn(sd, pre + "ffn_depth", od))                          # :272
    cls_feat = F.relu(_ln(sd, pre + "cls_fcs.1", _lin(sd, pre + "cls_fcs.0", o, bias=False)))     # :278-279
    mask_feat = F.relu(_ln(sd, pre + "mask_fcs.1", _lin(sd, pre + "mask_fcs.0", o, bias=False)))  # :280-281
    dep_feat = _ln(sd, pre + "depth_regs.1", _lin(sd, pre + "depth_regs.0", od, bias=False))      # :282-283 (no act)
    cls = _lin(sd, pre + "fc_cls", cls_feat)                         # :285
    kmask = _lin(sd, pre + "fc_mask", mask_feat)                     # :287
    kdep = _lin(sd, pre + "fc_depth", dep_feat)                      # :288
    new_m = torch.einsum("bnc,bchw->bnhw", kmask, xt)                # :317-322 (1x1 dynamic conv)
    new_d = torch.einsum("bnc,bchw->bnhw", kdep, dt)                 # :323-329
    return dict(cls=cls, mask=new_m, obj=o, depth=new_d, dobj=od, pooled=u, pooled_depth=ud,
                kmask=kmask, kdep=kdep)


def upsample2x(t, s=2):
    """kernel_update.py:131-143: bilinear, align_corners=False, scale_factor = mask_upsample_stride."""
    return F.interpolate(t, scale_factor=s, mode="bilinear", align_corners=False)


# --------------------------------------------------------------------------------------------
# A.4  KernelUpdateIterHead.simple_test_mask_preds (polyphonic/kernel_update.py:356-401)
# --------------------------------------------------------------------------------------------
def iter_head_mask_preds(sd, S, x, k0, m0, q0, dfe, heads=8, prefix="mask_head.", upsample=2,
                         return_stages=False, hard_masks=None):
    """k0/q0 accept [B,N,C,1,1] or [B,N,C] (q0 may be a stride-0 expand view).
    Returns dict(obj [B,N,C], cls sigmoid [B,N,L], mask [B,N,H,W], mask_up, depth, depth_up, dobj).
    `hard_masks` (tests only): list of S entries, None or the {0,1} mask stage s pools with (see update_stage)."""
    B, N = k0.shape[:2]
    k = k0.reshape(B, N, -1)
    q = q0.reshape(B, N, -1)
    m = m0
    stages = []
    for s in range(S):                                               # :383-394
        r = update_stage(sd, f"{prefix}{s}.", x, k, m, q, dfe, heads,
                         hard_mask=None if hard_masks is None else hard_masks[s])
        k, q, m = r["obj"], r["dobj"], r["mask"]
        if return_stages:
            stages.append(r)
    out = dict(obj=k, dobj=q, cls=r["cls"].sigmoid(), cls_logits=r["cls"], mask=m,
               depth=r["depth"],
               mask_up=upsample2x(m, upsample), depth_up=upsample2x(r["depth"], upsample))  # :131-143, :396-397
    if return_stages:
        out["stages"] = stages
    return out


# --------------------------------------------------------------------------------------------
# A.1  KernelHead._decode_init_proposals, post-neck part (polyphonic/kernel_head.py:245-347)
# --------------------------------------------------------------------------------------------
def _conv_gn_relu(sd, name, f, groups):
    y = F.conv2d(f, sd[name + ".conv.weight"])                       # ConvModule: bias absent (norm present)
    y = F.group_norm(y, groups, sd[name + ".gn.weight"], sd[name + ".gn.bias"], GN_EPS)
    return F.relu(y)


def kernel_head_post_neck(sd, f0, f1, f2, num_thing_classes, num_classes, groups=32, prefix="",
                          cat_stuff_mask=True, hard_mask=None):
    """`hard_mask` (tests only): {0,1} [B, Nth, H, W] used INSTEAD of binarize(m_th) in the object pooling (see update_stage)."""
    p = prefix
    B = f0.shape[0]
    loc = _conv_gn_relu(sd, p + "loc_convs.0", f0, groups)           # :250-251
    sem = _conv_gn_relu(sd, p + "seg_convs.0", f1, groups)           # :264-265
    dfe = _conv_gn_relu(sd, p + "depth_convs.0", f2, groups)         # :277-278
    W_init = sd[p + "init_kernels.weight"]                           # [Nth, C, 1, 1]
    m_th = F.conv2d(loc, W_init)                                     # :256
    dpr = F.conv2d(dfe, sd[p + "conv_direct_depth.weight"], sd[p + "conv_direct_depth.bias"])  # :285
    seg = F.conv2d(sem, sd[p + "conv_seg.weight"], sd[p + "conv_seg.bias"])                    # :295
    x = sem + loc                                                    # :303
    Mth = binarize(m_th) if hard_mask is None else hard_mask.to(m_th.dtype)
    obj = torch.einsum("bnhw,bchw->bnc", Mth, x)                     # :314-320 (use_binary)
    Nth, C = W_init.shape[:2]
    k0 = W_init[None].expand(B, Nth, C, 1, 1) + obj.view(B, Nth, C, 1, 1)   # :299-300,324-326
    dker = sd[p + "conv_direct_depth.weight"][None].expand(B, 1, C, 1, 1)    # :286-289
    mask_preds, proposal = m_th, k0
    depth_proposal = dker
    if cat_stuff_mask:                                               # :329-336 (eval)
        mask_preds = torch.cat([m_th, seg[:, num_thing_classes:num_classes]], dim=1)
        stuff_k = sd[p + "conv_seg.weight"][num_thing_classes:num_classes]
        proposal = torch.cat([k0, stuff_k[None].expand(B, *stuff_k.shape)], dim=1)
        depth_proposal = dker.expand(-1, proposal.shape[1], -1, -1, -1)
    return dict(proposal_feats=proposal, x_feats=x, mask_preds=mask_preds, seg_preds=seg,
                depth_feats=dfe, depth_proposal=depth_proposal, depth_pred=dpr)


# --------------------------------------------------------------------------------------------
# A.5  panoptic merge (polyphonic/kernel_update.py:421-535, kernel_update_head.py:593-626)
# --------------------------------------------------------------------------------------------
def rescale(t, img_meta):
    """kernel_update_head.py:593-608 / :610-626 (t already activated): bilinear to
    batch_input_shape, crop to img_shape, bilinear to ori_shape. t: [K, h, w]."""
    h, w = img_meta["img_shape"][:2]
    t = F.interpolate(t[None], size=tuple(img_meta["batch_input_shape"]), mode="bilinear",
                      align_corners=False)
    t = t[:, :, :h, :w]
    t = F.interpolate(t, size=tuple(img_meta["ori_shape"][:2]), mode="bilinear", align_corners=False)
    return t[0]


def select_segments(cls_scores, num_proposals, num_thing_classes, max_per_img):
    """kernel_update.py:428-434 (things top-k) and :448-459 (stuff diagonal, sorted).
    cls_scores: [N, L] post-sigmoid.  Returns (query_index[K], label[K], score[K]) with the
    things first, then the stuff, as `merge_stuff_thing_stuff_joint` concatenates them (:487-489)."""
    thing = cls_scores[:num_proposals][:, :num_thing_classes]
    tscore, tidx = thing.flatten(0, 1).topk(max_per_img, sorted=True)
    tq = tidx // num_thing_classes
    tl = tidx % num_thing_classes
    sscore = cls_scores[num_proposals:][:, num_thing_classes:].diag()
    sscore, sind = torch.sort(sscore, descending=True)
    q = torch.cat([tq, sind + num_proposals])
    lab = torch.cat([tl, sind + num_thing_classes])
    sc = torch.cat([tscore, sscore])
    return q, lab, sc


def merge_from_probs(P, D, scores, labels, D0, num_thing_classes, instance_score_thr=0.3,
                     overlap_thr=0.6, order=None):
    """kernel_update.py:484-535 on materialised maps.  P [K,H,W] probabilities, D [K,H,W] depths,
    D0 [H,W] initial depth.  `order` overrides argsort(-scores) (torch.argsort is not stable; the
    caller can pin the permutation the reference used)."""
    K, H, W = P.shape
    pan = torch.zeros((H, W), dtype=torch.int32)
    ids = (scores.view(-1, 1, 1) * P).argmax(0)                      # :492-494
    if order is None:
        order = torch.argsort(-scores)                               # :497
    depth_final = D0.clone()
    seg_id = 0
    info = []
    for kk in order.tolist():                                        # :500
        cls = int(labels[kk])
        isthing = cls < num_thing_classes
        if isthing and bool(scores[kk] < instance_score_thr):        # :503 (0-dim fp32 tensor vs Python scalar: an fp32 comparison)
            continue
        mask = ids == kk                                             # :506
        area = int(mask.sum())
        orig = int((P[kk] >= 0.5).sum())                             # :508
        if area > 0 and orig > 0:
            if area / orig < overlap_thr:                            # :511
                continue
            seg_id += 1
            pan[mask] = seg_id                                       # :515
            depth_final[mask] = D[kk][mask]                          # :517
            if isthing:
                info.append(dict(id=seg_id, isthing=True, score=float(scores[kk]),
                                 category_id=cls, instance_id=kk))
            else:
                info.append(dict(id=seg_id, isthing=False, category_id=cls, area=area))
    return pan.numpy(), info, depth_final


def get_panoptic(cls_scores, mask_up, depth_up, depth_init_up, img_meta, num_proposals,
                 num_thing_classes, max_per_img, instance_score_thr=0.3, overlap_thr=0.6,
                 depth_mode="sigmoid"):
    """kernel_update.py:421-469 for one image.  mask_up/depth_up [N,2H,2W] logits,
    depth_init_up [1,2H,2W] logits (x2-upsampled `depth_pred`, :302-307)."""
    q, lab, sc = select_segments(cls_scores, num_proposals, num_thing_classes, max_per_img)
    P = rescale(mask_up[q].sigmoid(), img_meta)                      # :435,452 via rescale_masks
    D = rescale(depth_act(depth_up, depth_mode), img_meta)[q]        # :423,439-440,456
    D0 = rescale(depth_act(depth_init_up, depth_mode), img_meta)[0]  # :424,445
    pan, info, dfinal = merge_from_probs(P, D, sc, lab, D0, num_thing_classes,
                                         instance_score_thr, overlap_thr)
    return pan, info, D0.numpy(), dfinal.numpy()


# --------------------------------------------------------------------------------------------
# whole path, as Polyphonic.simple_test wires it (polyphonic/polyphonic_former.py:145-161)
# --------------------------------------------------------------------------------------------
def run_head(sd, feats, S, num_thing_classes, num_classes, heads=8, groups=32, hard_masks=None):
    """`hard_masks` (tests only): the S hard masks [B, N, H, W] of another implementation's run; stage s pools with
    hard_masks[s], KernelHead's object pooling with the thing rows of hard_masks[0] (the same binarisation of m_th)."""
    nth = sd["rpn_head.init_kernels.weight"].shape[0]
    kh = kernel_head_post_neck(sd, feats[0], feats[1], feats[2], num_thing_classes, num_classes,
                               groups, prefix="rpn_head.",
                               hard_mask=None if hard_masks is None else hard_masks[0][:, :nth])
    out = iter_head_mask_preds(sd, S, kh["x_feats"], kh["proposal_feats"], kh["mask_preds"],
                               kh["depth_proposal"], kh["depth_feats"], heads,
                               prefix="roi_head.mask_head.", hard_masks=hard_masks)
    out["kernel_head"] = kh
    return out


def to_numpy_tree(o):
    if isinstance(o, torch.Tensor):
        return o.detach().cpu().numpy()
    if isinstance(o, dict):
        return {k: to_numpy_tree(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [to_numpy_tree(v) for v in o]
    return o
