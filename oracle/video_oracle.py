"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the video association inputs:
mask -> box, RoI level mapping, RoIAlign, track embedding head.  Same rules as oracle/poly_oracle.py.

Pinning: `mask_stat_boxes`, `mask_extent_boxes` and `track_embed_head` are checked against the reference's own
functions/classes run in the build container (tests/golden/video.npz, written by oracle/gen_golden.py).
`roi_align` restates mmcv-full 1.3.18's RoIAlign(output_size=7, sampling_ratio=2, pool_mode='avg', aligned=True)
(the un-vendored dependency pinned in scripts/docker_env/Dockerfile:11-12; call site
configs/polyphonic_video/poly_r50_cityscapes_1x.py:65-71 -> mmdet SingleRoIExtractor ->
polyphonic_former_video.py:416).  mmcv is not installable here, so for that one op parity is UNPINNED by a reference
run: it is anchored on the published algorithm (Mask R-CNN RoIAlign with the half-pixel `aligned` offset) and on
analytic properties tested in tests/test_video_oracle.py (constant / affine fields)."""
import math

import numpy as np
import torch
import torch.nn.functional as F


def mask_stat_boxes(masks):
    """polyphonic/video/utils.py:61-83 (batch_mask2boxlist via coords2bboxTensor, extend=2) for one image.
    masks [n,H,W] bool/float.  Returns xyxy boxes [n,4]: centre +- 2 * mean absolute deviation per axis."""
    out = []
    for m in masks:
        c = m.nonzero().float()                        # (row, col)
        if c.numel() == 0:
            out.append(torch.zeros(4))
            continue
        ctr = c.mean(0)
        dx = max(torch.sqrt((c[:, 0] - ctr[0]) ** 2).mean(), torch.tensor(1.0))    # rows  ("x" in the reference)
        dy = max(torch.sqrt((c[:, 1] - ctr[1]) ** 2).mean(), torch.tensor(1.0))    # cols
        left, right = ctr[0] - dx * 2, ctr[0] + dx * 2
        top, bottom = ctr[1] - dy * 2, ctr[1] + dy * 2
        out.append(torch.stack([top, left, bottom, right]))      # :57 -> [x1, y1, x2, y2] with x = col, y = row
    return torch.stack(out) if out else torch.zeros((0, 4))


def mask_extent_boxes(masks):
    """polyphonic/funcs/utils.py:4-22 (tensor_mask2box): tight extents, (-1,-1,10,10) for an empty mask."""
    out = []
    for m in masks:
        c = m.nonzero().float()
        if c.numel() == 0:
            out.append(torch.tensor([-1., -1., 10., 10.]))
        else:
            out.append(torch.stack([c[:, 1].min(), c[:, 0].min(), c[:, 1].max(), c[:, 0].max()]))
    return torch.stack(out) if out else torch.zeros((0, 4))


def map_roi_levels(rois, num_levels=4, finest_scale=56):
    """mmdet SingleRoIExtractor.map_roi_levels (single_level_roi_extractor.py:36-56); rois [k,5]."""
    scale = torch.sqrt((rois[:, 3] - rois[:, 1]) * (rois[:, 4] - rois[:, 2]))
    lv = torch.floor(torch.log2(scale / finest_scale + 1e-6))
    return lv.clamp(min=0, max=num_levels - 1).long()


def _bilinear(feat, y, x):
    """feat [C,H,W]; y, x scalars (python floats).  mmcv roi_align bilinear_interpolate."""
    C, H, W = feat.shape
    if y < -1.0 or y > H or x < -1.0 or x > W:
        return torch.zeros(C)
    y, x = max(y, 0.0), max(x, 0.0)
    yl, xl = int(y), int(x)
    if yl >= H - 1:
        yh = yl = H - 1
        y = float(yl)
    else:
        yh = yl + 1
    if xl >= W - 1:
        xh = xl = W - 1
        x = float(xl)
    else:
        xh = xl + 1
    ly, lx = y - yl, x - xl
    hy, hx = 1.0 - ly, 1.0 - lx
    return hy * hx * feat[:, yl, xl] + hy * lx * feat[:, yl, xh] + ly * hx * feat[:, yh, xl] + ly * lx * feat[:, yh, xh]


def roi_align(feat, rois, spatial_scale, out_size=7, sampling_ratio=2, aligned=True):
    """feat [1,C,H,W] fp32, rois [k,5] (batch index ignored: one image).  Returns [k,C,out,out]."""
    f = feat[0]
    k = rois.shape[0]
    out = torch.zeros((k, f.shape[0], out_size, out_size))
    off = 0.5 if aligned else 0.0
    for i in range(k):
        x1, y1, x2, y2 = [float(np.float32(v) * np.float32(spatial_scale)) - off for v in rois[i, 1:].tolist()]
        rw, rh = x2 - x1, y2 - y1
        if not aligned:
            rw, rh = max(rw, 1.0), max(rh, 1.0)
        bw, bh = rw / out_size, rh / out_size
        g = sampling_ratio
        for ph in range(out_size):
            for pw in range(out_size):
                acc = torch.zeros(f.shape[0])
                for iy in range(g):
                    yy = y1 + ph * bh + (iy + 0.5) * bh / g
                    for ix in range(g):
                        xx = x1 + pw * bw + (ix + 0.5) * bw / g
                        acc += _bilinear(f, yy, xx)
                out[i, :, ph, pw] = acc / (g * g)
    return out


def roi_extract(feats, rois, strides=(4, 8, 16, 32), finest_scale=56):
    """SingleRoIExtractor.forward (single_level_roi_extractor.py:58-113), one image."""
    lv = map_roi_levels(rois, len(strides), finest_scale)
    out = torch.zeros((rois.shape[0], feats[0].shape[1], 7, 7))
    for l, s in enumerate(strides):
        idx = (lv == l).nonzero().squeeze(1)
        if idx.numel():
            out[idx] = roi_align(feats[l], rois[idx], 1.0 / s)
    return out


def track_embed_head(sd, x, groups=32, prefix="track_head."):
    """QuasiDenseMaskEmbedHeadGTMask.forward (polyphonic/video/track_heads.py:92-102):
    4 x (conv3x3 no-bias -> GN(32) -> ReLU), flatten NCHW, fc -> ReLU, fc_embed.  x [n,256,7,7]."""
    for i in range(4):
        x = F.conv2d(x, sd[f"{prefix}convs.{i}.conv.weight"], padding=1)
        x = F.relu(F.group_norm(x, groups, sd[f"{prefix}convs.{i}.gn.weight"], sd[f"{prefix}convs.{i}.gn.bias"], 1e-5))
    x = x.reshape(x.shape[0], -1)
    x = F.relu(F.linear(x, sd[prefix + "fcs.0.weight"], sd[prefix + "fcs.0.bias"]))
    return F.linear(x, sd[prefix + "fc_embed.weight"], sd[prefix + "fc_embed.bias"])


def things_for_tracking(pan, info):
    """PolyphonicVideo.get_things_id_for_tracking (polyphonic_former_video.py:421-434)"""
    idxs, labels, masks, score = [], [], [], []
    for s in info:
        if s["isthing"]:
            masks.append(torch.from_numpy(pan == s["id"]))
            idxs.append(s["instance_id"])
            labels.append(s["category_id"])
            score.append(s["score"])
    return idxs, labels, masks, score
