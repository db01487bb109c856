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


# ---- QuasiDenseEmbedTracker (polyphonic/video/qdtrack/trackers/quasi_dense_embed_tracker.py:8-207) ---------------------------------
# Restated for the tests that compare the product's association END TO END (tests/test_gpu_video.py): plain torch CPU ops, one
# dictionary of tracklets + a list of backdrop frames, no velocity (the reference computes one and never reads it).  Pinned by
# tests/golden/tracker.npz -- ids / labels / boxes the reference class produced on three synthetic clips -- in
# tests/test_video_oracle.py::test_tracker_oracle_reproduces_the_reference_ids.
def box_iou(a, b, eps=1e-6):
    """mmdet.core.bbox_overlaps(mode='iou', is_aligned=False): [n, 4] x [m, 4] -> [n, m]"""
    if a.shape[0] == 0 or b.shape[0] == 0:
        return a.new_zeros((a.shape[0], b.shape[0]))
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = torch.max(a[:, None, :2], b[None, :, :2])
    rb = torch.min(a[:, None, 2:4], b[None, :, 2:4])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    union = torch.max(area_a[:, None] + area_b[None, :] - inter, inter.new_tensor(eps))
    return inter / union


class TrackerOracle:
    def __init__(self, init_score_thr=0.8, obj_score_thr=0.5, match_score_thr=0.5, memo_tracklet_frames=10, memo_backdrop_frames=1,
                 memo_momentum=0.8, nms_conf_thr=0.5, nms_backdrop_iou_thr=0.3, nms_class_iou_thr=0.7, with_cats=True,
                 match_metric="bisoftmax"):
        self.p = dict(init=init_score_thr, obj=obj_score_thr, match=match_score_thr, keep=memo_tracklet_frames, bd_frames=memo_backdrop_frames,
                      mom=memo_momentum, conf=nms_conf_thr, bd_iou=nms_backdrop_iou_thr, cls_iou=nms_class_iou_thr)
        self.with_cats, self.metric = with_cats, match_metric
        self.tracks = {}            # id -> dict(embed, label, last) in creation order (:106-118 walks the dict in that order)
        self.backdrops = []         # newest first: dict(embeds, labels)
        self.born = 0

    def _memory(self):
        """:100-133: tracklet columns in creation order, then the backdrop frames newest first with id -1"""
        ids = list(self.tracks)
        emb = [self.tracks[i]["embed"][None] for i in ids]
        lab = [self.tracks[i]["label"].reshape(1) for i in ids]
        for b in self.backdrops:
            ids += [-1] * b["embeds"].shape[0]
            emb.append(b["embeds"])
            lab.append(b["labels"])
        return torch.tensor(ids, dtype=torch.long), torch.cat(emb, 0), torch.cat(lab, 0)

    def match(self, bboxes, labels, track_feats, frame_id):
        p = self.p
        order = bboxes[:, -1].sort(descending=True)[1]                                              # :137-140
        box, lab, emb = bboxes[order], labels[order], track_feats[order]
        iou = box_iou(box[:, :4], box[:, :4])
        keep = torch.ones(box.shape[0], dtype=torch.bool)                                           # :144-152
        for i in range(1, box.shape[0]):
            thr = p["bd_iou"] if box[i, -1] < p["obj"] else p["cls_iou"]
            if (iou[i, :i] > thr).any():
                keep[i] = False
        box, lab, emb = box[keep], lab[keep], emb[keep]
        n = box.shape[0]
        ids = torch.full((n,), -1, dtype=torch.long)
        if n and self.tracks:                                                                       # :161 (`empty` looks at the tracklets only)
            m_ids, m_emb, m_lab = self._memory()
            if self.metric == "cosine":
                score = F.normalize(emb, p=2, dim=1) @ F.normalize(m_emb, p=2, dim=1).t()
            else:
                dots = emb @ m_emb.t()
                score = dots.softmax(1)
                if self.metric == "bisoftmax":
                    score = (score + dots.softmax(0)) / 2
            if self.with_cats:
                score = score * (lab[:, None] == m_lab[None, :]).float()
            for i in range(n):                                                                      # :183-197
                conf, j = score[i].max(0)
                if conf > p["match"] and m_ids[j] > -1:
                    if box[i, -1] > p["obj"]:
                        ids[i] = m_ids[j]
                        score[:i, j] = 0
                        score[i + 1:, j] = 0
                    elif conf > p["conf"]:
                        ids[i] = -2
        new = (ids == -1) & (box[:, 4] > p["init"])                                                 # :198-205
        k = int(new.sum())
        ids[new] = torch.arange(self.born, self.born + k, dtype=torch.long)
        self.born += k
        # update_memo (:47-98)
        for i in range(n):
            t = int(ids[i])
            if t < 0:
                continue
            if t in self.tracks:
                tr = self.tracks[t]
                tr["embed"] = (1 - p["mom"]) * tr["embed"] + p["mom"] * emb[i]
                tr["label"], tr["last"] = lab[i], frame_id
            else:
                self.tracks[t] = dict(embed=emb[i], label=lab[i], last=frame_id)
        loose = torch.nonzero(ids == -1).squeeze(1)
        iou2 = box_iou(box[loose, :4], box[:, :4])
        free = [int(ind) for r, ind in enumerate(loose) if not (iou2[r, :int(ind)] > p["bd_iou"]).any()]
        self.backdrops.insert(0, dict(embeds=emb[free], labels=lab[free]))
        for t in [t for t, tr in self.tracks.items() if frame_id - tr["last"] >= p["keep"]]:
            del self.tracks[t]
        if len(self.backdrops) > p["bd_frames"]:
            self.backdrops.pop()
        return box, lab, ids
