"""Video association step (the consumer of the one multi-GPU exchange):
drop-in for `QuasiDenseEmbedTracker` (polyphonic/video/qdtrack/trackers/quasi_dense_embed_tracker.py:8-207).

The tracker is *host logic*: stateful, strictly sequential in frame order, data-dependent control flow over
<= max_per_img detections and a memo of a few hundred rows (the reference runs it as ~10 tiny torch ops plus
Python loops with `.item()`-style syncs).  It is mirrored here as Python over CPU tensors, same constructor
kwargs, same `match(bboxes, labels, track_feats, frame_id)` signature and return value, same registry name.
With frames sharded over GPUs (`dist.shard_frames`) every rank all-gathers the per-frame records
(`dist.allgather_track_records`) and replays `match` in frame order; integer track ids are then identical to the
single-process run (`replay_tracking`, tests/test_tracker.py and tests/test_dist_gloo.py)."""
import torch

from .registry import Registry

TRACKERS = Registry("trackers")


def bbox_overlaps(b1, b2, eps=1e-6):
    """IoU matrix, mmdet.core.bbox_overlaps(mode='iou', is_aligned=False) for xyxy boxes [n,4] x [m,4]"""
    if b1.shape[0] == 0 or b2.shape[0] == 0:
        return b1.new_zeros((b1.shape[0], b2.shape[0]))
    area1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    area2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    lt = torch.max(b1[:, None, :2], b2[None, :, :2])
    rb = torch.min(b1[:, None, 2:], b2[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    overlap = wh[..., 0] * wh[..., 1]
    union = area1[:, None] + area2[None, :] - overlap
    union = torch.max(union, union.new_tensor([eps]))
    return overlap / union


@TRACKERS.register_module()
class QuasiDenseEmbedTracker(object):

    def __init__(self, init_score_thr=0.8, obj_score_thr=0.5, match_score_thr=0.5, memo_tracklet_frames=10,
                 memo_backdrop_frames=1, memo_momentum=0.8, nms_conf_thr=0.5, nms_backdrop_iou_thr=0.3,
                 nms_class_iou_thr=0.7, with_cats=True, match_metric='bisoftmax'):
        assert 0 <= memo_momentum <= 1.0
        assert memo_tracklet_frames >= 0
        assert memo_backdrop_frames >= 0
        self.init_score_thr, self.obj_score_thr, self.match_score_thr = init_score_thr, obj_score_thr, match_score_thr
        self.memo_tracklet_frames, self.memo_backdrop_frames = memo_tracklet_frames, memo_backdrop_frames
        self.memo_momentum, self.nms_conf_thr = memo_momentum, nms_conf_thr
        self.nms_backdrop_iou_thr, self.nms_class_iou_thr, self.with_cats = nms_backdrop_iou_thr, nms_class_iou_thr, with_cats
        assert match_metric in ['bisoftmax', 'softmax', 'cosine']
        self.match_metric = match_metric
        self.num_tracklets = 0
        self.tracklets = dict()
        self.backdrops = []

    @property
    def empty(self):
        return False if self.tracklets else True

    def update_memo(self, ids, bboxes, embeds, labels, frame_id):
        """quasi_dense_embed_tracker.py:47-102"""
        tracklet_inds = ids > -1
        for id, bbox, embed, label in zip(ids[tracklet_inds], bboxes[tracklet_inds], embeds[tracklet_inds],
                                          labels[tracklet_inds]):
            id = int(id)
            if id in self.tracklets:
                t = self.tracklets[id]
                velocity = (bbox - t['bbox']) / (frame_id - t['last_frame'])
                t['bbox'] = bbox
                t['embed'] = (1 - self.memo_momentum) * t['embed'] + self.memo_momentum * embed
                t['last_frame'] = frame_id
                t['label'] = label
                t['velocity'] = (t['velocity'] * t['acc_frame'] + velocity) / (t['acc_frame'] + 1)
                t['acc_frame'] += 1
            else:
                self.tracklets[id] = dict(bbox=bbox, embed=embed, label=label, last_frame=frame_id,
                                          velocity=torch.zeros_like(bbox), acc_frame=0)
        backdrop_inds = torch.nonzero(ids == -1, as_tuple=False).squeeze(1)
        ious = bbox_overlaps(bboxes[backdrop_inds, :-1], bboxes[:, :-1])
        for i, ind in enumerate(backdrop_inds):
            if (ious[i, :ind] > self.nms_backdrop_iou_thr).any():
                backdrop_inds[i] = -1
        backdrop_inds = backdrop_inds[backdrop_inds > -1]
        self.backdrops.insert(0, dict(bboxes=bboxes[backdrop_inds], embeds=embeds[backdrop_inds],
                                      labels=labels[backdrop_inds]))
        invalid_ids = [k for k, v in self.tracklets.items() if frame_id - v['last_frame'] >= self.memo_tracklet_frames]
        for k in invalid_ids:
            self.tracklets.pop(k)
        if len(self.backdrops) > self.memo_backdrop_frames:
            self.backdrops.pop()

    @property
    def memo(self):
        """quasi_dense_embed_tracker.py:104-135"""
        memo_embeds, memo_ids, memo_bboxes, memo_labels, memo_vs = [], [], [], [], []
        for k, v in self.tracklets.items():
            memo_bboxes.append(v['bbox'][None, :])
            memo_embeds.append(v['embed'][None, :])
            memo_ids.append(k)
            memo_labels.append(v['label'].view(1, 1))
            memo_vs.append(v['velocity'][None, :])
        memo_ids = torch.tensor(memo_ids, dtype=torch.long).view(1, -1)
        for backdrop in self.backdrops:
            backdrop_ids = torch.full((1, backdrop['embeds'].size(0)), -1, dtype=torch.long)
            memo_bboxes.append(backdrop['bboxes'])
            memo_embeds.append(backdrop['embeds'])
            memo_ids = torch.cat([memo_ids, backdrop_ids], dim=1)
            memo_labels.append(backdrop['labels'][:, None])
            memo_vs.append(torch.zeros_like(backdrop['bboxes']))
        return (torch.cat(memo_bboxes, dim=0), torch.cat(memo_labels, dim=0).squeeze(1), torch.cat(memo_embeds, dim=0),
                memo_ids.squeeze(0), torch.cat(memo_vs, dim=0))

    def match(self, bboxes, labels, track_feats, frame_id, asso_tau=-1):
        """quasi_dense_embed_tracker.py:137-207.  bboxes [n,5] (x1,y1,x2,y2,score), labels [n], track_feats [n,256]."""
        bboxes, labels, track_feats = bboxes.detach().cpu().float(), labels.detach().cpu().long(), track_feats.detach().cpu().float()
        _, inds = bboxes[:, -1].sort(descending=True)
        bboxes, labels, embeds = bboxes[inds, :], labels[inds], track_feats[inds, :]
        valids = bboxes.new_ones((bboxes.size(0)))
        ious = bbox_overlaps(bboxes[:, :-1], bboxes[:, :-1])
        for i in range(1, bboxes.size(0)):
            thr = self.nms_backdrop_iou_thr if bboxes[i, -1] < self.obj_score_thr else self.nms_class_iou_thr
            if (ious[i, :i] > thr).any():
                valids[i] = 0
        valids = valids == 1
        bboxes, labels, embeds = bboxes[valids, :], labels[valids], embeds[valids, :]
        ids = torch.full((bboxes.size(0),), -1, dtype=torch.long)
        if bboxes.size(0) > 0 and not self.empty:
            memo_bboxes, memo_labels, memo_embeds, memo_ids, memo_vs = self.memo
            if self.match_metric == 'bisoftmax':
                feats = torch.mm(embeds, memo_embeds.t())
                scores = (feats.softmax(dim=1) + feats.softmax(dim=0)) / 2
            elif self.match_metric == 'softmax':
                scores = torch.mm(embeds, memo_embeds.t()).softmax(dim=1)
            else:
                scores = torch.mm(torch.nn.functional.normalize(embeds, p=2, dim=1),
                                  torch.nn.functional.normalize(memo_embeds, p=2, dim=1).t())
            if self.with_cats:
                scores *= (labels.view(-1, 1) == memo_labels.view(1, -1)).float()
            for i in range(bboxes.size(0)):
                conf, memo_ind = torch.max(scores[i, :], dim=0)
                id = memo_ids[memo_ind]
                if conf > self.match_score_thr:
                    if id > -1:
                        if bboxes[i, -1] > self.obj_score_thr:
                            ids[i] = id
                            scores[:i, memo_ind] = 0
                            scores[i + 1:, memo_ind] = 0
                        else:
                            if conf > self.nms_conf_thr:
                                ids[i] = -2
        new_inds = (ids == -1) & (bboxes[:, 4] > self.init_score_thr)
        num_news = int(new_inds.sum())
        ids[new_inds] = torch.arange(self.num_tracklets, self.num_tracklets + num_news, dtype=torch.long)
        self.num_tracklets += num_news
        self.update_memo(ids, bboxes, embeds, labels, frame_id)
        return bboxes, labels, ids


def replay_tracking(records, tracker_cfg=None, tracker=None):
    """records: [(frame_id, bboxes[n,5], labels[n], embeds[n,256])] of ALL frames (any order); replays `match` in
    frame order like polyphonic_former_video.py:391-402 (frame_id = running count from 1, ids + 1, -1 -> 0).
    Returns {frame_id: ids tensor}."""
    if tracker is None:
        tracker = QuasiDenseEmbedTracker(**(tracker_cfg or {}))
    out, cnt = {}, 1
    for fid, bb, lab, emb in sorted(records, key=lambda r: r[0]):
        if bb.shape[0] > 0:
            _, _, ids = tracker.match(bboxes=bb, labels=lab, track_feats=emb, frame_id=cnt)
            cnt += 1
            ids = ids + 1
            ids[ids == -1] = 0
        else:
            ids = torch.zeros((0,), dtype=torch.long)
        out[fid] = ids
    return out


# ---- the per-frame association step of PolyphonicVideo.simple_test (polyphonic_former_video.py:359-405) ----------
INSTANCE_DIVISOR = 10000       # datasets/cityscapes_dvps.py (max_ins): pred_pan = sem * 10000 + track_id


def things_for_tracking(panoptic_seg, segments_info):
    """get_things_id_for_tracking (:421-434): segment ids, labels and scores of the thing segments, in segment order"""
    seg_ids, idxs, labels, score = [], [], [], []
    for s in segments_info:
        if s['isthing']:
            seg_ids.append(s['id'])
            idxs.append(s['instance_id'])
            labels.append(s['category_id'])
            score.append(s['score'])
    return seg_ids, idxs, labels, score


def semantic_map(panoptic_seg, segments_info, num_thing_classes, num_stuff_classes):
    """get_semantic_seg (:436-440) as one table lookup; void = num_thing + num_stuff (uint8 like the reference)"""
    import numpy as np
    lut = np.full(int(panoptic_seg.max()) + 1, num_thing_classes + num_stuff_classes, dtype=np.uint8)
    for s in segments_info:
        lut[s['id']] = s['category_id']
    return lut[panoptic_seg]


def track_id_map(panoptic_seg, seg_ids, ids):
    """generate_track_id_maps (:442-451): float64 map, 0 = no track (the masks are `panoptic_seg == segment id`)"""
    import numpy as np
    lut = np.zeros(int(panoptic_seg.max()) + 1, dtype=np.float64)
    for sid, tid in zip(seg_ids, ids):
        lut[sid] = float(tid)
    return lut[panoptic_seg]


def wire_record(result):
    """datasets/cityscapes_dvps.py:325-338 (pre_eval): what is saved per frame for DVPQ evaluation (see dvps_eval.py)"""
    from .dvps_eval import wire_record as _w
    return _w(result)


class VideoAssociator:
    """What PolyphonicVideo.simple_test does after `roi_head.simple_test` (:359-405), for one video stream:
    boxes from the id map -> FPN RoIAlign -> track embeddings (libpolyhead) -> tracker -> sem / track / depth maps.
    `records_only=True` returns the (bboxes, labels, embeds) record instead of matching, for the sharded mode where
    the records are all-gathered and replayed in frame order (`dist.allgather_track_records`, `replay_tracking`)."""

    def __init__(self, track_head, tracker_cfg, num_thing_classes, num_stuff_classes, strides=(4, 8, 16, 32)):
        self.track_head, self.tracker_cfg = track_head, dict(tracker_cfg)
        self.num_thing_classes, self.num_stuff_classes, self.strides = num_thing_classes, num_stuff_classes, strides
        self.init_tracker()

    def init_tracker(self):
        """polyphonic_former_video.py:59-61"""
        self.tracker = QuasiDenseEmbedTracker(**self.tracker_cfg)
        self.cnt = 1

    def record(self, fpn_feats, panoptic_seg, segments_info):
        from . import track_head as T, engine as E
        seg_ids, idxs, labels, score = things_for_tracking(panoptic_seg, segments_info)
        if not seg_ids:
            return seg_ids, None
        dev = fpn_feats[0].device
        rois_all, ext_all = T.segment_boxes(torch.from_numpy(panoptic_seg).to(dev), int(max(s['id'] for s in segments_info)))
        sel = torch.tensor([i - 1 for i in seg_ids], device=dev)
        prec = E.PREC[self.track_head.precision]
        embeds = self.track_head.forward_planes(T.roi_extract(fpn_feats, rois_all[sel].contiguous(), prec, self.strides))
        bboxes = torch.cat([ext_all[sel], torch.tensor(score, device=dev, dtype=torch.float32)[:, None]], 1)
        return seg_ids, (bboxes.cpu(), torch.tensor(labels, dtype=torch.int64), embeds.cpu())

    def step(self, fpn_feats, panoptic_seg, segments_info, depth_final, records_only=False):
        seg_ids, rec = self.record(fpn_feats, panoptic_seg, segments_info)
        if records_only:
            return seg_ids, rec
        ids = []
        if rec is not None:
            # NB the reference sorts detections by score inside `match`; ids come back in that order (:142-145) and are
            # painted onto the masks in segment order (:403) -- mirrored as is
            _, _, ids = self.tracker.match(bboxes=rec[0], labels=rec[1], track_feats=rec[2], frame_id=self.cnt)
            self.cnt += 1
            ids = ids + 1
            ids[ids == -1] = 0
            ids = ids.tolist()
        return [{"sem": semantic_map(panoptic_seg, segments_info, self.num_thing_classes, self.num_stuff_classes),
                 "track": track_id_map(panoptic_seg, seg_ids, ids), "depth": depth_final}]
