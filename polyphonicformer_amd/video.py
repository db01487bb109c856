"""Video association step (the consumer of the one multi-GPU exchange):
drop-in for `QuasiDenseEmbedTracker` (polyphonic/video/qdtrack/trackers/quasi_dense_embed_tracker.py:8-207).

The tracker is *host logic*: stateful, strictly sequential in frame order, data-dependent control flow over
<= max_per_img detections and a memory of a few hundred rows.  This build keeps that memory as a struct of arrays
(`_TrackTable`) and formulates de-duplication, affinity and greedy assignment as array operations; constructor
kwargs, the `match(bboxes, labels, track_feats, frame_id)` signature / return value and the registry name are the
reference's, and the integer ids are pinned to the reference class's own output (tests/golden/tracker.npz).
With frames sharded over GPUs (`dist.shard_frames`) every rank all-gathers the per-frame records
(`dist.allgather_track_records`) and replays `match` in frame order; integer track ids are then identical to the
single-process run (`replay_tracking`, tests/test_tracker.py and tests/test_dist_gloo.py)."""
import torch

from .registry import Registry

TRACKERS = Registry("trackers")


def _idx_to(dev, idx):
    """a host index tensor to the device the embeddings live on: through pinned memory and asynchronously when that is a GPU
    (six such transfers per frame in `match`; pageable ones cost 30-100 us each on the GPU box), a no-op on the CPU"""
    if torch.device(dev).type != "cuda":
        return idx
    return idx.pin_memory().to(dev, non_blocking=True)


def bbox_overlaps(b1, b2, eps=1e-6):
    """IoU matrix of xyxy boxes [n,4] x [m,4] (what mmdet.core.bbox_overlaps(mode='iou') returns)"""
    n, m = b1.shape[0], b2.shape[0]
    if n == 0 or m == 0:
        return b1.new_zeros((n, m))
    x1 = torch.maximum(b1[:, None, 0], b2[None, :, 0])
    y1 = torch.maximum(b1[:, None, 1], b2[None, :, 1])
    x2 = torch.minimum(b1[:, None, 2], b2[None, :, 2])
    y2 = torch.minimum(b1[:, None, 3], b2[None, :, 3])
    inter = (x2 - x1).clamp(min=0) * (y2 - y1).clamp(min=0)
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    return inter / (a1[:, None] + a2[None, :] - inter).clamp(min=eps)


class _TrackTable:
    """The tracker's memory as a struct of arrays: one row per live tracklet, in creation order (the order the
    affinity columns are laid out in, which decides ties), plus the most recent frames' unmatched detections
    ("backdrops", newest frame first).  Rows are updated / appended / expired with index operations; nothing here is
    per-object Python state."""

    def __init__(self, backdrop_frames):
        self.ids = torch.zeros((0,), dtype=torch.long)
        self.box = torch.zeros((0, 5))
        self.emb = torch.zeros((0, 0))                            # lives where the track head left its embeddings (device)
        self.lab = torch.zeros((0,), dtype=torch.long)
        self.seen = torch.zeros((0,), dtype=torch.long)          # frame a row was last matched in
        self.backdrop_frames = backdrop_frames
        self.backdrops = []                                       # [(box, emb, lab)], newest first

    def __len__(self):
        return int(self.ids.numel())

    def columns(self):
        """(ids, labels, embeds) of everything a detection can be matched to: tracklets, then backdrops (id -1)"""
        ids, lab, emb = [self.ids], [self.lab], [self.emb]
        for (bb, be, bl) in self.backdrops:
            ids.append(torch.full((be.shape[0],), -1, dtype=torch.long))
            lab.append(bl)
            emb.append(be)
        emb = [e for e in emb if e.shape[0]]
        return torch.cat(ids), torch.cat(lab), (torch.cat(emb, 0) if emb else self.emb)

    def absorb(self, ids, box, emb, lab, frame, momentum):
        """matched detections refresh their rows (embedding = exponential moving average), unknown ids append rows"""
        if ids.numel() == 0:
            return
        pos = {int(t): r for r, t in enumerate(self.ids.tolist())}
        row = torch.tensor([pos.get(int(t), -1) for t in ids.tolist()], dtype=torch.long)
        old, new = row >= 0, row < 0
        dev = emb.device
        sel = lambda mask: _idx_to(dev, mask.nonzero().flatten())   # index tensors (host masks) for the device-resident embeddings
        if old.any():
            r = row[old]
            rd = _idx_to(dev, r)
            self.emb[rd] = (1 - momentum) * self.emb[rd] + momentum * emb[sel(old)]
            self.box[r], self.lab[r], self.seen[r] = box[old], lab[old], frame
        if new.any():
            k = int(new.sum())
            self.ids = torch.cat([self.ids, ids[new]])
            self.box = torch.cat([self.box, box[new]], 0)
            self.emb = torch.cat([self.emb.reshape(-1, emb.shape[1]).to(dev), emb[sel(new)]], 0)
            self.lab = torch.cat([self.lab, lab[new]])
            self.seen = torch.cat([self.seen, torch.full((k,), frame, dtype=torch.long)])

    def expire(self, frame, max_age):
        live = (frame - self.seen) < max_age
        if not bool(live.all()):
            self.ids, self.box, self.emb, self.lab, self.seen = (self.ids[live], self.box[live],
                                                                  self.emb[_idx_to(self.emb.device, live.nonzero().flatten())],
                                                                  self.lab[live], self.seen[live])

    def push_backdrop(self, box, emb, lab):
        self.backdrops.insert(0, (box, emb, lab))
        del self.backdrops[self.backdrop_frames:]
        if self.backdrop_frames == 0:
            self.backdrops = []


@TRACKERS.register_module()
class QuasiDenseEmbedTracker(object):
    """Quasi-dense embedding tracker with the constructor kwargs, `match` signature and integer-id semantics of
    polyphonic/video/qdtrack/trackers/quasi_dense_embed_tracker.py:8-207 (pinned by tests/golden/tracker.npz, which the
    reference class produced).  This build's formulation: detections are de-duplicated with one triangular IoU test,
    the memory is a `_TrackTable`, the affinity matrix is computed once and the greedy one-to-one assignment walks the
    detections in score order with a `taken` mask over tracklet columns.  The reference also carries a per-tracklet
    velocity that nothing reads (its `match` ignores `memo_vs`); it is not kept.
    Where the arithmetic runs: boxes, labels, ids and the control flow on the host; the EMBEDDINGS (detections and memory)
    stay on the device the track head produced them on, and the [detections x memory] affinity matrix is computed there
    (`ph_track_affinity`, csrc/ph_track.hip) -- one D2H of that matrix per frame feeds the greedy walk.  With CPU inputs
    (the CPU tests, gloo) the same arithmetic runs as torch CPU ops; no process-global state is touched either way."""

    def __init__(self, init_score_thr=0.8, obj_score_thr=0.5, match_score_thr=0.5, memo_tracklet_frames=10,
                 memo_backdrop_frames=1, memo_momentum=0.8, nms_conf_thr=0.5, nms_backdrop_iou_thr=0.3,
                 nms_class_iou_thr=0.7, with_cats=True, match_metric='bisoftmax'):
        if not (0 <= memo_momentum <= 1.0) or memo_tracklet_frames < 0 or memo_backdrop_frames < 0:
            raise AssertionError("bad memo configuration")
        if match_metric not in ('bisoftmax', 'softmax', 'cosine'):
            raise AssertionError(f"unknown match_metric {match_metric}")
        self.init_score_thr, self.obj_score_thr, self.match_score_thr = init_score_thr, obj_score_thr, match_score_thr
        self.memo_tracklet_frames, self.memo_backdrop_frames = memo_tracklet_frames, memo_backdrop_frames
        self.memo_momentum, self.nms_conf_thr = memo_momentum, nms_conf_thr
        self.nms_backdrop_iou_thr, self.nms_class_iou_thr, self.with_cats = nms_backdrop_iou_thr, nms_class_iou_thr, with_cats
        self.match_metric = match_metric
        self.num_tracklets = 0
        self.table = _TrackTable(memo_backdrop_frames)

    @property
    def empty(self):
        return len(self.table) == 0

    # -- pieces of `match` ---------------------------------------------------------------------------------
    def _dedup(self, box):
        """a detection is dropped when ANY higher-scored detection (kept or not) overlaps it by more than the IoU
        threshold of its own score class (:147-155)"""
        iou = bbox_overlaps(box[:, :4], box[:, :4])
        thr = torch.where(box[:, 4] < self.obj_score_thr, torch.tensor(self.nms_backdrop_iou_thr),
                          torch.tensor(self.nms_class_iou_thr))
        return ~(torch.tril(iou, -1) > thr[:, None]).any(1), iou

    def _affinity(self, emb, lab, memo_emb, memo_lab):
        """[detections x memory columns] match scores (:165-182), returned on the host"""
        # the fused kernel keeps a detection row in one workgroup: n <= 128 detections, m <= 4096 memory columns (max_per_img is
        # 100 and the memory a few hundred columns in the shipped configs).  Beyond that the same formula runs as torch ops ON
        # THE DEVICE the embeddings live on (below) -- the reference has no limit, a long video must not abort mid-stream
        if emb.is_cuda and emb.shape[0] <= 128 and memo_emb.shape[0] <= 4096:
            from . import _lib
            lib = _lib.load()
            n, m = emb.shape[0], memo_emb.shape[0]
            dev = emb.device
            score = torch.empty((n, m), dtype=torch.float32, device=dev)
            ws = torch.empty((lib.ph_track_affinity_workspace_bytes(n, m),), dtype=torch.uint8, device=dev)
            metric = {'bisoftmax': 0, 'softmax': 1, 'cosine': 2}[self.match_metric]
            # named, so that the four operands are alive (and distinct blocks of the caching allocator) until the launch is queued
            e, me = emb.contiguous(), memo_emb.contiguous()
            l, ml = lab.to(dev, torch.int32), memo_lab.to(dev, torch.int32)
            _lib.check(lib.ph_track_affinity(_lib.ptr(e), _lib.ptr(l), _lib.ptr(me), _lib.ptr(ml), n, m, metric, 1 if self.with_cats else 0,
                                             _lib.ptr(score), _lib.ptr(ws), ws.numel(), _lib.stream_ptr()), "ph_track_affinity")
            return score.cpu()
        if self.match_metric == 'cosine':
            unit = torch.nn.functional.normalize
            s = unit(emb, p=2, dim=1) @ unit(memo_emb, p=2, dim=1).t()
        else:
            dot = emb @ memo_emb.t()
            s = dot.softmax(dim=1)
            if self.match_metric == 'bisoftmax':
                s = (s + dot.softmax(dim=0)) / 2
        if self.with_cats:
            s = s * (lab.to(s.device)[:, None] == memo_lab.to(s.device)[None, :]).float()
        return s.cpu()

    def _assign(self, score, det_conf, memo_ids):
        """greedy, in detection (score) order: best still-free column; a tracklet column is consumed by a confident
        detection, a weak detection that resembles a tracklet is marked -2 (neither a new track nor a backdrop),
        matches to backdrop columns assign nothing (:183-197)"""
        n = score.shape[0]
        out = torch.full((n,), -1, dtype=torch.long)
        taken = torch.zeros(score.shape[1], dtype=torch.bool)
        for i in range(n):
            conf, j = score[i].masked_fill(taken, 0).max(0)
            if not conf > self.match_score_thr or memo_ids[j] < 0:
                continue
            if det_conf[i] > self.obj_score_thr:
                out[i] = memo_ids[j]
                taken[j] = True
            elif conf > self.nms_conf_thr:
                out[i] = -2
        return out

    def match(self, bboxes, labels, track_feats, frame_id, asso_tau=-1):
        """bboxes [n,5] (x1,y1,x2,y2,score), labels [n], track_feats [n,256] -> (bboxes, labels, ids) of the kept
        detections in descending-score order; ids >= 0 track, -1 unmatched, -2 suppressed."""
        return self._match(bboxes, labels, track_feats, frame_id)

    def _match(self, bboxes, labels, track_feats, frame_id):
        box, lab, emb = bboxes.detach().cpu().float(), labels.detach().cpu().long(), track_feats.detach().float()   # emb: stays put
        dev = emb.device
        order = box[:, 4].sort(descending=True)[1]
        keep, _ = self._dedup(box[order])
        kept = order[keep]                                        # one gather of the embeddings for both steps
        box, lab, emb = box[kept], lab[kept], emb[_idx_to(dev, kept)]
        ids = torch.full((box.shape[0],), -1, dtype=torch.long)
        if box.shape[0] and not self.empty:
            memo_ids, memo_lab, memo_emb = self.table.columns()
            ids = self._assign(self._affinity(emb, lab, memo_emb, memo_lab), box[:, 4], memo_ids)
        born = (ids == -1) & (box[:, 4] > self.init_score_thr)
        k = int(born.sum())
        ids[born] = torch.arange(self.num_tracklets, self.num_tracklets + k, dtype=torch.long)
        self.num_tracklets += k
        self._remember(ids, box, emb, lab, frame_id)
        return box, lab, ids

    def _remember(self, ids, box, emb, lab, frame_id):
        """:47-102: tracked detections go to the table; the still-unmatched ones that no higher-scored detection covers
        become this frame's backdrops; tracklets unseen for `memo_tracklet_frames` frames are forgotten"""
        tracked = ids > -1
        self.table.absorb(ids[tracked], box[tracked], emb[_idx_to(emb.device, tracked.nonzero().flatten())], lab[tracked], frame_id,
                          self.memo_momentum)
        loose = ids == -1
        iou = bbox_overlaps(box[:, :4], box[:, :4])
        covered = (torch.tril(iou, -1) > self.nms_backdrop_iou_thr).any(1)
        bd = loose & ~covered
        self.table.push_backdrop(box[bd], emb[_idx_to(emb.device, bd.nonzero().flatten())], lab[bd])
        self.table.expire(frame_id, self.memo_tracklet_frames)


def replay_tracking(records, tracker_cfg=None, tracker=None, first_count=1):
    """records: [(frame_id, bboxes[n,5], labels[n], embeds[n,256])] of ALL frames (any order); replays `match` in
    frame order like polyphonic_former_video.py:391-402 (frame_id = running count from 1, ids + 1, -1 -> 0).
    A stream that arrives in batches passes its persistent `tracker` and `first_count` = 1 + the number of non-empty frames
    replayed so far.  Returns {frame_id: ids tensor}."""
    if tracker is None:
        tracker = QuasiDenseEmbedTracker(**(tracker_cfg or {}))
    out, cnt = {}, first_count
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

    def record(self, fpn_feats, panoptic_seg, segments_info, pan_dev=None):
        from . import track_head as T, engine as E
        seg_ids, idxs, labels, score = things_for_tracking(panoptic_seg, segments_info)
        if not seg_ids:
            return seg_ids, None
        dev = fpn_feats[0].device
        if pan_dev is None:
            pan_dev = torch.from_numpy(panoptic_seg).to(dev)
        rois_all, ext_all = T.segment_boxes(pan_dev, int(max(s['id'] for s in segments_info)))
        sel = _h2d([i - 1 for i in seg_ids], torch.int64, dev)
        prec = E.PREC[self.track_head.precision]
        embeds = self.track_head.forward_planes(T.roi_extract(fpn_feats, rois_all[sel].contiguous(), prec, self.strides))
        bboxes = torch.cat([ext_all[sel], _h2d(score, torch.float32, dev)[:, None]], 1)
        return seg_ids, (bboxes.cpu(), torch.tensor(labels, dtype=torch.int64), embeds)      # the embeddings stay on the device

    def _maps_on_device(self, pan_dev, segments_info, seg_ids, ids, to_host=True):
        """get_semantic_seg / generate_track_id_maps (:436-451) as two table look-ups on the device copy of the id map
        (the host versions `semantic_map` / `track_id_map` walk 2 M pixels in numpy: 10 ms per 1024x2048 frame)"""
        n = int(max([s['id'] for s in segments_info], default=0)) + 1
        sem_lut = torch.full((n,), self.num_thing_classes + self.num_stuff_classes, dtype=torch.uint8)
        trk_lut = torch.zeros((n,), dtype=torch.float64)
        for s in segments_info:
            sem_lut[s['id']] = s['category_id']
        for sid, tid in zip(seg_ids, ids):
            trk_lut[sid] = float(tid)
        idx = pan_dev.long()
        sem = sem_lut.pin_memory().to(pan_dev.device, non_blocking=True)[idx]
        trk = trk_lut.pin_memory().to(pan_dev.device, non_blocking=True)[idx]
        if not to_host:
            return sem, trk
        return sem.cpu().numpy(), trk.cpu().numpy()

    def step_device(self, fpn_feats, pan_dev, segments_info):
        """`step` on the DEVICE copy of the panoptic id map, results left on the device: (sem uint8, track float64) maps.
        Same kernels, same tracker calls, same values as `step` -- for callers that overlap the result download with the
        next frame (`VideoStreamRunner`)."""
        seg_ids, rec = self.record(fpn_feats, None, segments_info, pan_dev)
        ids = []
        if rec is not None:
            _, _, ids = self.tracker.match(bboxes=rec[0], labels=rec[1], track_feats=rec[2], frame_id=self.cnt)
            self.cnt += 1
            ids = ids + 1
            ids[ids == -1] = 0
            ids = ids.tolist()
        return self._maps_on_device(pan_dev, segments_info, seg_ids, ids, to_host=False)

    def step(self, fpn_feats, panoptic_seg, segments_info, depth_final, records_only=False):
        pan_dev = torch.from_numpy(panoptic_seg).to(fpn_feats[0].device)
        seg_ids, rec = self.record(fpn_feats, panoptic_seg, segments_info, pan_dev)
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
        sem, trk = self._maps_on_device(pan_dev, segments_info, seg_ids, ids)
        return [{"sem": sem, "track": trk, "depth": depth_final}]


class VideoFramePipeline:
    """`PolyphonicVideo.simple_test` after `extract_feat` (polyphonic_former_video.py:327-405), in the reference's order:

        rpn_head.simple_test_rpn(x, img_metas)                      -> proposals, post-neck maps, mask / depth logits
        roi_head.simple_test(...)                                    -> panoptic_seg, segments_info, depth_final  (a6 + a7)
        get_things_id_for_tracking -> boxes / RoIAlign on the FPN levels `x` -> track_head -> tracker.match
        -> [{"sem": uint8 map, "track": float64 map, "depth": fp32 map}]

    One instance per video stream (the tracker is stateful; `init_tracker` starts a new video, :59-61).  Backbone and FPN
    are the caller's (`x` = the four FPN levels of ONE frame, as `extract_feat` returns them)."""

    def __init__(self, rpn_head, roi_head, track_head, tracker_cfg, strides=(4, 8, 16, 32)):
        self.rpn_head, self.roi_head = rpn_head, roi_head
        self.assoc = VideoAssociator(track_head, tracker_cfg, roi_head.num_thing_classes, roi_head.num_stuff_classes, strides)

    def init_tracker(self):
        self.assoc.init_tracker()

    def heads(self, x, img_metas, rescale=False):
        """the two heads exactly as :343-357 calls them; returns roi_head.simple_test's result list"""
        (proposal_feats, x_feats, mask_preds, cls_scores, seg_preds, depth_feats, depth_proposal, depth_pred,
         semantic_aspp_out) = self.rpn_head.simple_test_rpn(x, img_metas)
        return self.roi_head.simple_test(x_feats, proposal_feats, mask_preds, cls_scores, img_metas, depth_preds=depth_pred,
                                         depth_feats=depth_feats, depth_proposal=depth_proposal, imgs_whwh=None,
                                         aspp_semantic=semantic_aspp_out, rescale=rescale)

    def simple_test(self, x, img_metas, rescale=False, records_only=False):
        if x[0].shape[0] != 1:
            raise NotImplementedError("video inference is one frame at a time (samples_per_gpu = 1, as in the reference)")
        _, _, (panoptic_seg, segments_info), _, depth_final = self.heads(x, img_metas, rescale)[0]
        return self.assoc.step(x, panoptic_seg, segments_info, depth_final, records_only=records_only)


def _h2d(values, dtype, dev):
    """a small host list -> device tensor through pinned memory, asynchronously (torch.tensor(..., device=dev) goes through
    pageable memory and costs ~110 us per call on the GPU box: six of them per frame)"""
    return torch.tensor(values, dtype=dtype).pin_memory().to(dev, non_blocking=True)


class VideoStreamRunner:
    """Throughput form of the reference's per-frame video loop (polyphonic/apis/video_inference.py:8-31 ->
    PolyphonicVideo.simple_test, polyphonic_former_video.py:327-405) for ONE stream of equally sized frames: the same kernels,
    the same tracker calls and therefore the same results as `VideoFramePipeline.simple_test`, issued so that the GPU box's
    host is not the bottleneck (round 4; the module-API loop spends ~80 % of a 5 ms frame on the host):

      * neck -> KernelHead -> 3-stage decode -> x2 upsample of depth_pred are ONE HIP graph per slot, captured on first use from
        the unmodified module calls (`rpn_head.simple_test_rpn`, `roi_head._decode`) on static copies of the FPN levels and
        replayed for every later frame (~70 kernel launches -> one graph launch);
      * TWO slots (the second one a deep copy of the two head modules with its own plans and buffers): the heads of frame t run
        on their slot's stream while the host walks frame t - 1 through merge -> boxes -> RoIAlign -> track head -> tracker,
        whose four small D2H reads (class scores, area histograms, boxes, affinity matrix) synchronise the MAIN stream only.
        The tracker sees the frames in order; heads are frame-independent;
      * the panoptic id map never visits the host: the merge's result stays on the device (`panoptic.get_panoptic_device`)
        where the association step consumes it;
      * what the reference returns as numpy -- the uint8 semantic map, the float64 track-id map, the fp32 depth map (26 MB per
        1024x2048 frame) -- is copied to pinned host memory on a side stream under the following frames.

    `push(x)` therefore returns the result of frame t - 2 (None for the first two frames); `flush()` returns the list of the
    results still in flight, oldest first.  `pipelined=False`: one slot, results one frame late (round 4's first form).
    Weights are packed at capture time: call `reset()` after changing them."""

    def __init__(self, pipe, img_meta, graph=True, pipelined=True):
        self.pipe, self.metas, self.use_graph, self.pipelined = pipe, [img_meta], graph, pipelined
        self.reset()

    def reset(self):
        self._slots = []                 # per slot: dict(rpn, roi, x, graph, outs, stream, done)
        self._copy_stream = None
        self._inflight = None            # frame whose heads are running: (slot index)
        self._downloads = []             # [(event, host tensors, device sources)] oldest first
        self._n = 0

    # -- slots ---------------------------------------------------------------------------------------------------
    def _slot(self, i):
        while len(self._slots) <= i:
            if not self._slots:
                rpn, roi = self.pipe.rpn_head, self.pipe.roi_head
            else:
                rpn, roi = self._clone_heads()
            self._slots.append(dict(rpn=rpn, roi=roi, x=None, graph=None, outs=None, stream=torch.cuda.Stream(), done=None))
        return self._slots[i]

    def _clone_heads(self):
        """a second pair of head modules with the same weights and its own plans / buffers (the originals' plans -- GBs of
        device buffers, HIP streams -- are taken out while the modules are copied)"""
        import copy
        rpn, roi = self.pipe.rpn_head, self.pipe.roi_head
        held = [(m, m._plans) for m in (rpn, roi, getattr(rpn, "localization_fpn", None)) if m is not None and hasattr(m, "_plans")]
        for m, _ in held:
            m._plans = {}
        try:
            rpn2, roi2 = copy.deepcopy(rpn), copy.deepcopy(roi)
        finally:
            for m, pl in held:
                m._plans = pl
        return rpn2, roi2

    def _heads_device(self, sl, x):
        from . import engine as E
        (proposal_feats, x_feats, mask_preds, cls_scores, seg_preds, depth_feats, depth_proposal, depth_pred,
         semantic_aspp_out) = sl["rpn"].simple_test_rpn(x, self.metas)
        o = sl["roi"]._decode(x_feats, proposal_feats, mask_preds, depth_feats, depth_proposal)
        depth_init = E.upsample2x(depth_pred.float().contiguous())                         # kernel_update.py:302-307
        return o["cls"], o["mask_up"], o["depth_up"], depth_init

    def _start_heads(self, i, x):
        """copy the frame into slot i's static inputs and start its heads on the slot's stream"""
        sl = self._slot(i)
        main = torch.cuda.current_stream()
        if not self.use_graph:
            sl["x"] = x
            sl["outs"] = self._heads_device(sl, x)
            sl["done"] = torch.cuda.Event()
            sl["done"].record(main)
            return
        if sl["graph"] is None:
            sl["x"] = tuple(torch.empty_like(t) for t in x)
            for d, t in zip(sl["x"], x):
                d.copy_(t)
            self._heads_device(sl, sl["x"])                 # warm-up outside the capture: plans, packs, kernel attributes
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                sl["outs"] = self._heads_device(sl, sl["x"])
            sl["graph"] = g
        if len(x) != len(sl["x"]) or any(tuple(d.shape) != tuple(t.shape) or d.dtype != t.dtype for d, t in zip(sl["x"], x)):
            raise ValueError("VideoStreamRunner: the FPN levels changed shape / dtype; one runner serves one stream of equally "
                             "sized frames (call reset() to re-capture)")
        for d, t in zip(sl["x"], x):
            d.copy_(t, non_blocking=True)                    # on the caller's stream: x may be reused once push returns
        ready = torch.cuda.Event()
        ready.record(main)
        with torch.cuda.stream(sl["stream"]):
            sl["stream"].wait_event(ready)
            sl["graph"].replay()
            sl["done"] = torch.cuda.Event()
            sl["done"].record(sl["stream"])

    # -- the part of a frame that follows the heads -----------------------------------------------------------------
    def _merge(self, i):
        from . import panoptic as Pn
        sl = self._slots[i]
        torch.cuda.current_stream().wait_event(sl["done"])
        cls, mask_up, depth_up, depth_init = sl["outs"]
        return Pn.get_panoptic_device(sl["roi"], cls[0], mask_up[0], depth_up[0], depth_init[0], self.metas[0])

    def _finish(self, i):
        """merge -> association -> start the download of frame (slot i)'s result maps"""
        pan_dev, info, _, d_final = self._merge(i)
        sem, trk = self.pipe.assoc.step_device(self._slots[i]["x"], pan_dev, info)
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream()
        main = torch.cuda.current_stream()
        done_main = torch.cuda.Event()
        done_main.record(main)
        # fresh pinned buffers (the caller owns them); the device sources stay referenced until the copy has finished
        host = [torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in (sem, trk, d_final)]
        with torch.cuda.stream(self._copy_stream):
            self._copy_stream.wait_event(done_main)
            for h, t in zip(host, (sem, trk, d_final)):
                t.record_stream(self._copy_stream)
                h.copy_(t, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._copy_stream)
        self._downloads.append((ev, host, (sem, trk, d_final)))

    @staticmethod
    def _collect(p):
        ev, host, _keep = p
        ev.synchronize()
        return [{"sem": host[0].numpy(), "track": host[1].numpy(), "depth": host[2].numpy()}]

    def _check(self, x):
        if x[0].shape[0] != 1:
            raise NotImplementedError("video inference is one frame at a time (samples_per_gpu = 1, as in the reference)")

    def push(self, x):
        """x: the four FPN levels of ONE frame (device tensors).  Returns the result list [{"sem", "track", "depth"}] (numpy,
        owned by the caller) of the frame pushed two calls ago (one call ago with pipelined=False), or None."""
        self._check(x)
        if not self.pipelined:
            self._start_heads(0, x)
            self._finish(0)
            self._n += 1
            return self._collect(self._downloads.pop(0)) if len(self._downloads) > 1 else None
        i = self._n & 1
        self._start_heads(i, x)                              # frame t: heads on slot i's stream ...
        if self._inflight is not None:
            self._finish(self._inflight)                     # ... while the host takes frame t - 1 through merge / association
        self._inflight = i
        self._n += 1
        return self._collect(self._downloads.pop(0)) if len(self._downloads) > 1 else None

    def flush(self):
        """the results still in flight, oldest first (a list of result lists)"""
        if self._inflight is not None:
            self._finish(self._inflight)
            self._inflight = None
        out = [self._collect(p) for p in self._downloads]
        self._downloads = []
        return out

    def records(self, frames):
        """the sharded mode's per-step work for a rank's clip: `simple_test(..., records_only=True)` of every frame in order.
        The heads of up to two frames are in flight at a time: frame k + 1's are started BEFORE frame k's merge / record, and the
        clip's first two frames start back to back.  Returns [(segment ids, (bboxes, labels, embeds) or None)] per frame."""
        frames = list(frames)
        out, started = [], 0
        assert self._inflight is None, "records() and push() / push_record() must not be interleaved"
        depth = 2 if self.pipelined else 1
        for k in range(len(frames)):
            while started < len(frames) and started < k + depth:
                self._check(frames[started])
                self._start_heads(started % depth, frames[started])
                started += 1
            out.append(self._record(k % depth))
        return out

    def push_record(self, x):
        """the sharded mode's per-frame work (`simple_test(..., records_only=True)`) for the frame pushed ONE call ago (None for
        the first call): its heads ran while the caller dealt with the frame before; `flush_record()` returns the last frame's.
        Returns (segment ids, (bboxes, labels, embeds) or None); nothing map-sized is downloaded."""
        self._check(x)
        i = self._n & 1 if self.pipelined else 0
        prev = None
        if not self.pipelined:
            self._start_heads(0, x)
            self._n += 1
            return self._record(0)
        self._start_heads(i, x)
        if self._inflight is not None:
            prev = self._record(self._inflight)
        self._inflight = i
        self._n += 1
        return prev

    def flush_record(self):
        if self._inflight is None:
            return None
        i, self._inflight = self._inflight, None
        return self._record(i)

    def _record(self, i):
        pan_dev, info, _, _ = self._merge(i)
        return self.pipe.assoc.record(self._slots[i]["x"], None, info, pan_dev)
