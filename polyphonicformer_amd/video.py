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
import os

import numpy as np
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


def _iou_np(b1, b2, eps=1e-6):
    """`bbox_overlaps` on numpy float32 arrays: the same fp32 operations in the same order (bit-identical values), at numpy's
    per-call cost -- the tracker's bookkeeping is ~60 tiny array operations per frame"""
    n, m = b1.shape[0], b2.shape[0]
    if n == 0 or m == 0:
        return np.zeros((n, m), dtype=np.float32)
    x1 = np.maximum(b1[:, None, 0], b2[None, :, 0])
    y1 = np.maximum(b1[:, None, 1], b2[None, :, 1])
    x2 = np.minimum(b1[:, None, 2], b2[None, :, 2])
    y2 = np.minimum(b1[:, None, 3], b2[None, :, 3])
    inter = np.maximum(x2 - x1, np.float32(0)) * np.maximum(y2 - y1, np.float32(0))
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    return inter / np.maximum(a1[:, None] + a2[None, :] - inter, np.float32(eps))


def _rows_to(dev, rows):
    """numpy row indices -> an index tensor where the embeddings live"""
    return _idx_to(dev, torch.from_numpy(np.ascontiguousarray(rows, dtype=np.int64)))


class _TrackTable:
    """The tracker's memory as a struct of arrays: one row per live tracklet, in creation order (the order the
    affinity columns are laid out in, which decides ties), plus the most recent frames' unmatched detections
    ("backdrops", newest frame first).  Rows are updated / appended / expired with index operations; nothing here is
    per-object Python state.  ids / boxes / labels / last-seen frames are numpy arrays on the host, the embeddings a torch
    tensor on the device the track head left them on."""

    def __init__(self, backdrop_frames):
        self.ids = np.zeros((0,), dtype=np.int64)
        self.box = np.zeros((0, 5), dtype=np.float32)
        self.emb = torch.zeros((0, 0))                            # lives where the track head left its embeddings (device)
        self.lab = np.zeros((0,), dtype=np.int64)
        self.seen = np.zeros((0,), dtype=np.int64)                # frame a row was last matched in
        self.backdrop_frames = backdrop_frames
        self.backdrops = []                                       # [(box, emb, lab)], newest first

    def __len__(self):
        return int(self.ids.shape[0])

    def columns(self):
        """(ids, labels, embeds) of everything a detection can be matched to: tracklets, then backdrops (id -1)"""
        ids, lab, emb = [self.ids], [self.lab], [self.emb]
        for (bb, be, bl) in self.backdrops:
            ids.append(np.full((be.shape[0],), -1, dtype=np.int64))
            lab.append(bl)
            emb.append(be)
        emb = [e for e in emb if e.shape[0]]
        return np.concatenate(ids), np.concatenate(lab), (torch.cat(emb, 0) if len(emb) > 1 else (emb[0] if emb else self.emb))

    def absorb(self, ids, box, emb, lab, frame, momentum):
        """matched detections refresh their rows (embedding = exponential moving average), unknown ids append rows"""
        if ids.shape[0] == 0:
            return
        pos = {t: r for r, t in enumerate(self.ids.tolist())}
        row = np.array([pos.get(t, -1) for t in ids.tolist()], dtype=np.int64)
        old, new = row >= 0, row < 0
        dev = emb.device
        if old.any():
            r = row[old]
            rd = _rows_to(dev, r)
            self.emb[rd] = (1 - momentum) * self.emb[rd] + momentum * emb[_rows_to(dev, np.flatnonzero(old))]
            self.box[r], self.lab[r], self.seen[r] = box[old], lab[old], frame
        if new.any():
            k = int(new.sum())
            self.ids = np.concatenate([self.ids, ids[new]])
            self.box = np.concatenate([self.box, box[new]], 0)
            self.emb = torch.cat([self.emb.reshape(-1, emb.shape[1]).to(dev), emb[_rows_to(dev, np.flatnonzero(new))]], 0)
            self.lab = np.concatenate([self.lab, lab[new]])
            self.seen = np.concatenate([self.seen, np.full((k,), frame, dtype=np.int64)])

    def expire(self, frame, max_age):
        live = (frame - self.seen) < max_age
        if not live.all():
            self.ids, self.box, self.emb, self.lab, self.seen = (self.ids[live], self.box[live],
                                                                  self.emb[_rows_to(self.emb.device, np.flatnonzero(live))],
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
    Where the arithmetic runs: boxes, labels, ids and the control flow on the host, as numpy float32 / int64 arrays (the same
    IEEE operations as the torch CPU ops they replace -- same values, a third of the per-call cost; the frame's ~60 tiny array
    operations were 0.6 ms of a 2 ms video frame); the EMBEDDINGS (detections and memory) stay on the device the track head
    produced them on, and the [detections x memory] affinity matrix is computed there (`ph_track_affinity`, csrc/ph_track.hip)
    -- one D2H of that matrix per frame feeds the greedy walk.  With CPU inputs (the CPU tests, gloo) the affinity runs as
    torch CPU ops; no process-global state is touched either way."""

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
        self._num_tracklets = 0
        self.table = _TrackTable(memo_backdrop_frames)
        self._native = None           # (handle, device memory, device): round 5, embeddings on a GPU -> csrc/ph_tracker.hip

    @property
    def num_tracklets(self):
        if self._native is not None:
            from . import _lib
            return int(_lib.load().ph_tracker_num_tracklets(self._native[0]))
        return self._num_tracklets

    @num_tracklets.setter
    def num_tracklets(self, v):
        self._num_tracklets = v

    @property
    def empty(self):
        if self._native is not None:
            from . import _lib
            return _lib.load().ph_tracker_rows(self._native[0]) == 0
        return len(self.table) == 0

    def __del__(self):
        nat = getattr(self, "_native", None)
        if nat is not None:
            try:
                from . import _lib
                _lib.load().ph_tracker_destroy(nat[0])
            except Exception:
                pass

    # -- the native form (embeddings on a GPU): one C call per frame ----------------------------------------------
    NATIVE_CAPACITY, NATIVE_MAX_DETS = 4096, 128
    native = True                 # False: the array form below also for GPU embeddings (tests compare the two)

    def _native_handle(self, dev):
        """the C++ tracker object of this stream (csrc/ph_tracker.hip), created on first use on the embeddings' device"""
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        if self._native is None:
            if len(self.table) or self._num_tracklets:
                raise _lib.PolyheadError("QuasiDenseEmbedTracker: a tracker that started on CPU embeddings cannot continue on the GPU")

            class Cfg(C.Structure):
                _fields_ = [(n, C.c_float) for n in ("init_score_thr", "obj_score_thr", "match_score_thr", "memo_momentum", "one_minus_momentum",
                                                     "nms_conf_thr", "nms_backdrop_iou_thr", "nms_class_iou_thr")] + \
                           [(n, C.c_int32) for n in ("memo_tracklet_frames", "memo_backdrop_frames", "with_cats", "metric")]
            c = Cfg(self.init_score_thr, self.obj_score_thr, self.match_score_thr, self.memo_momentum, 1 - self.memo_momentum, self.nms_conf_thr,
                    self.nms_backdrop_iou_thr, self.nms_class_iou_thr, self.memo_tracklet_frames, self.memo_backdrop_frames,
                    1 if self.with_cats else 0, {'bisoftmax': 0, 'softmax': 1, 'cosine': 2}[self.match_metric])
            nb = lib.ph_tracker_device_bytes(self.NATIVE_CAPACITY, self.NATIVE_MAX_DETS)
            mem = torch.empty((nb,), dtype=torch.uint8, device=dev)
            h = lib.ph_tracker_create(C.byref(c), _lib.ptr(mem), nb, self.NATIVE_CAPACITY, self.NATIVE_MAX_DETS)
            if not h:
                raise _lib.PolyheadError("ph_tracker_create failed: " + (lib.ph_last_error_string() or b"").decode())
            self._native = (C.c_void_p(h), mem, dev)
        if dev != self._native[2]:
            raise _lib.PolyheadError("QuasiDenseEmbedTracker: the embeddings moved to another device mid-stream")
        return self._native[0]

    def native_ready(self, n, emb):
        """True if `match` of n detections with these embeddings takes the native path"""
        return bool(self.native and emb.is_cuda and n <= self.NATIVE_MAX_DETS and emb.shape[1] == 256
                    and (self._native is not None or (len(self.table) == 0 and self._num_tracklets == 0)))

    def match_frames(self, boxes, labels, embeds, first_frame_id):
        """a step's frames in ONE native call (`ph_tracker_match_frames`): boxes [sum n, 5] float32 / labels [sum n] int64 numpy arrays
        on the host, embeds: per frame a [n, 256] device tensor (frames without detections: n = 0, skipped like the reference's loop).
        Returns per frame the int64 ids of its kept detections in `match`'s order, and the number of frames matched."""
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        nf = len(embeds)
        counts = np.asarray([int(e.shape[0]) for e in embeds], dtype=np.int32)
        tot = int(counts.sum())
        dev = next((e.device for e in embeds if e.shape[0]), None)
        if dev is None:
            return [np.empty((0,), dtype=np.int64) for _ in embeds], 0
        h = self._native_handle(dev)
        embs = [e.detach().float().contiguous() for e in embeds]
        ptrs = (C.c_void_p * nf)(*[e.data_ptr() if e.shape[0] else None for e in embs])
        box = np.ascontiguousarray(boxes, dtype=np.float32)
        lab = np.ascontiguousarray(labels, dtype=np.int64)
        kept, ids, kc = np.empty((max(tot, 1),), dtype=np.int32), np.empty((max(tot, 1),), dtype=np.int64), np.empty((nf,), dtype=np.int32)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        m = lib.ph_tracker_match_frames(h, vp(box), vp(lab), ptrs, vp(counts), nf, int(first_frame_id), vp(kept), vp(ids), vp(kc), _lib.stream_ptr())
        if m < 0:
            _lib.check(m, "ph_tracker_match_frames")
        o = np.concatenate([[0], np.cumsum(counts)])
        return [ids[o[f]:o[f] + kc[f]].copy() for f in range(nf)], int(m)

    def _match_native(self, bboxes, labels, track_feats, frame_id):
        """csrc/ph_tracker.hip: bookkeeping in C++, embeddings in a device pool, per frame two small uploads, six launches and
        ONE synchronising download (the [detections x memory] scores).  Same integer ids as the array form below."""
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        dev = track_feats.device
        handle = self._native_handle(dev)
        # boxes / labels: torch tensors (any device) or host numpy arrays (`replay_tracking` downloads a whole step's at once)
        box = bboxes if isinstance(bboxes, np.ndarray) else bboxes.detach().cpu().float().numpy()
        lab = labels if isinstance(labels, np.ndarray) else labels.detach().cpu().long().numpy()
        box, lab = np.ascontiguousarray(box, dtype=np.float32), np.ascontiguousarray(lab, dtype=np.int64)
        emb = track_feats.detach().float().contiguous()
        n = box.shape[0]
        kept, ids = np.empty((max(n, 1),), dtype=np.int32), np.empty((max(n, 1),), dtype=np.int64)
        k = lib.ph_tracker_match(handle, box.ctypes.data_as(C.c_void_p), lab.ctypes.data_as(C.c_void_p), _lib.ptr(emb), n, int(frame_id),
                                 kept.ctypes.data_as(C.c_void_p), ids.ctypes.data_as(C.c_void_p), _lib.stream_ptr())
        if k < 0:
            _lib.check(k, "ph_tracker_match")
        kept = kept[:k].astype(np.int64)
        return torch.from_numpy(box[kept]), torch.from_numpy(lab[kept]), torch.from_numpy(ids[:k].copy())

    # -- pieces of `match` ---------------------------------------------------------------------------------
    def _dedup(self, box):
        """a detection is dropped when ANY higher-scored detection (kept or not) overlaps it by more than the IoU
        threshold of its own score class (:147-155).  box: numpy float32 [n, 5], descending score"""
        f32 = np.float32
        iou = _iou_np(box[:, :4], box[:, :4])
        thr = np.where(box[:, 4] < f32(self.obj_score_thr), f32(self.nms_backdrop_iou_thr), f32(self.nms_class_iou_thr))
        return ~(np.tril(iou, -1) > thr[:, None]).any(1), iou

    def _affinity(self, emb, lab, memo_emb, memo_lab):
        """[detections x memory columns] match scores (:165-182), returned on the host (torch fp32).  Labels: numpy or torch"""
        lab, memo_lab = torch.as_tensor(lab), torch.as_tensor(memo_lab)
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
            both = torch.cat([lab.reshape(-1), memo_lab.reshape(-1)]).to(torch.int32)          # ONE pinned transfer for both label vectors
            both = both.to(dev) if both.is_cuda else both.pin_memory().to(dev, non_blocking=True)
            l, ml = both[:n], both[n:]
            _lib.check(lib.ph_track_affinity(_lib.ptr(e), _lib.ptr(l), _lib.ptr(me), _lib.ptr(ml), n, m, metric, 1 if self.with_cats else 0,
                                             _lib.ptr(score), _lib.ptr(ws), ws.numel(), _lib.stream_ptr()), "ph_track_affinity")
            host = torch.empty((n, m), dtype=torch.float32, pin_memory=True)
            host.copy_(score, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            return host
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
        matches to backdrop columns assign nothing (:183-197).  numpy: score fp32 [n, m], det_conf fp32 [n], memo_ids int64 [m]"""
        f32 = np.float32
        n = score.shape[0]
        out = np.full((n,), -1, dtype=np.int64)
        taken = np.zeros(score.shape[1], dtype=bool)
        match_thr, obj_thr, conf_thr = f32(self.match_score_thr), f32(self.obj_score_thr), f32(self.nms_conf_thr)
        for i in range(n):
            row = np.where(taken, f32(0), score[i])
            j = int(row.argmax())                                  # first maximal column, like torch.max(0)
            conf = row[j]
            if not conf > match_thr or memo_ids[j] < 0:
                continue
            if det_conf[i] > obj_thr:
                out[i] = memo_ids[j]
                taken[j] = True
            elif conf > conf_thr:
                out[i] = -2
        return out

    def match(self, bboxes, labels, track_feats, frame_id, asso_tau=-1):
        """bboxes [n,5] (x1,y1,x2,y2,score), labels [n], track_feats [n,256] -> (bboxes, labels, ids) of the kept
        detections in descending-score order; ids >= 0 track, -1 unmatched, -2 suppressed."""
        return self._match(bboxes, labels, track_feats, frame_id)

    def _match(self, bboxes, labels, track_feats, frame_id):
        if self.native and track_feats.is_cuda and bboxes.shape[0] <= self.NATIVE_MAX_DETS and bboxes.shape[1] == 5 and track_feats.shape[1] == 256 \
                and (self._native is not None or (len(self.table) == 0 and self._num_tracklets == 0)):
            return self._match_native(bboxes, labels, track_feats, frame_id)
        if self._native is not None:
            from . import _lib
            # the native tracker owns the memory and the id counter from its first frame on: the array form below would start a second,
            # empty state (ids from 0 again, colliding with live native ids) and the native memory would miss this frame
            raise _lib.PolyheadError(f"QuasiDenseEmbedTracker: the native tracker has started and this frame does not fit it (device embeddings of "
                                f"width 256, [n, 5] boxes, n <= {self.NATIVE_MAX_DETS}; got n = {bboxes.shape[0]}, boxes {tuple(bboxes.shape)}, "
                                f"embeddings {tuple(track_feats.shape)} on {track_feats.device}); build the tracker with native=False for such streams")
        box_t, lab_t, emb = bboxes.detach().cpu().float(), labels.detach().cpu().long(), track_feats.detach().float()   # emb: stays put
        dev = emb.device
        order = box_t[:, 4].sort(descending=True)[1].numpy()      # torch's order among equal scores (what the goldens were pinned with)
        box, lab = box_t.numpy()[order], lab_t.numpy()[order]
        keep, iou = self._dedup(box)
        kept = order[keep]                                        # one gather of the embeddings for both steps
        box, lab, emb = box[keep], lab[keep], emb[_rows_to(dev, kept)]
        ids = np.full((box.shape[0],), -1, dtype=np.int64)
        if box.shape[0] and not self.empty:
            memo_ids, memo_lab, memo_emb = self.table.columns()
            ids = self._assign(self._affinity(emb, lab, memo_emb, memo_lab).numpy(), box[:, 4], memo_ids)
        born = (ids == -1) & (box[:, 4] > np.float32(self.init_score_thr))
        k = int(born.sum())
        ids[born] = np.arange(self._num_tracklets, self._num_tracklets + k, dtype=np.int64)
        self._num_tracklets += k
        self._remember(ids, box, emb, lab, frame_id, iou[keep][:, keep])
        return torch.from_numpy(box), torch.from_numpy(lab), torch.from_numpy(ids)

    def _remember(self, ids, box, emb, lab, frame_id, iou):
        """:47-102: tracked detections go to the table; the still-unmatched ones that no higher-scored detection covers
        become this frame's backdrops; tracklets unseen for `memo_tracklet_frames` frames are forgotten.  `iou`: the kept
        detections' pairwise IoU (the de-duplication's matrix restricted to them -- the same values)"""
        tracked = ids > -1
        self.table.absorb(ids[tracked], box[tracked], emb[_rows_to(emb.device, np.flatnonzero(tracked))], lab[tracked], frame_id,
                          self.memo_momentum)
        loose = ids == -1
        covered = (np.tril(iou, -1) > np.float32(self.nms_backdrop_iou_thr)).any(1)
        bd = loose & ~covered
        self.table.push_backdrop(box[bd], emb[_rows_to(emb.device, np.flatnonzero(bd))], lab[bd])
        self.table.expire(frame_id, self.memo_tracklet_frames)


def replay_tracking(records, tracker_cfg=None, tracker=None, first_count=1):
    """records: [(frame_id, bboxes[n,5], labels[n], embeds[n,256])] of ALL frames (any order); replays `match` in
    frame order like polyphonic_former_video.py:391-402 (frame_id = running count from 1, ids + 1, -1 -> 0).
    A stream that arrives in batches passes its persistent `tracker` and `first_count` = 1 + the number of non-empty frames
    replayed so far.  Returns {frame_id: ids tensor}."""
    if tracker is None:
        tracker = QuasiDenseEmbedTracker(**(tracker_cfg or {}))
    out, cnt = {}, first_count
    records = sorted(records, key=lambda r: r[0])
    if records and all(r[1].is_cuda and r[2].is_cuda for r in records):
        # device records (after an RCCL all-gather): ONE download of the step's boxes and labels instead of two per frame
        ns = [int(r[1].shape[0]) for r in records]
        if sum(ns):
            bl = torch.cat([torch.cat([r[1].float(), r[2].float()[:, None]], 1) for r in records if r[1].shape[0]], 0).cpu().numpy()
            o = np.cumsum([0] + ns)
            records = [(r[0], bl[o[i]:o[i + 1], :5], bl[o[i]:o[i + 1], 5].astype(np.int64), r[3]) for i, r in enumerate(records)]
    if records and all(isinstance(r[1], np.ndarray) and tracker.native_ready(r[1].shape[0], r[3]) for r in records):
        # device embeddings, host boxes: the whole step in ONE native call (a Python round trip per frame was half of the 0.13 ms a
        # frame's replay cost)
        per_frame, matched = tracker.match_frames(np.concatenate([r[1] for r in records], 0) if records else None,
                                                  np.concatenate([r[2] for r in records], 0), [r[3] for r in records], cnt)
        for r, ids in zip(records, per_frame):
            ids = ids + 1
            ids[ids == -1] = 0
            out[r[0]] = torch.from_numpy(ids)
        return out
    for fid, bb, lab, emb in records:
        if bb.shape[0] > 0:
            if isinstance(bb, np.ndarray) and not (tracker.native and emb.is_cuda and bb.shape[0] <= tracker.NATIVE_MAX_DETS):
                bb, lab = torch.from_numpy(bb), torch.from_numpy(lab)
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
        cfg = dict(self.tracker_cfg)
        cfg.setdefault("type", "QuasiDenseEmbedTracker")         # the config's own dict (with `type`) or bare kwargs
        self.tracker = TRACKERS.build(cfg)                       # build_tracker(self.tracker_cfg), polyphonic_former_video.py:60
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
        # the extent boxes start their way to the host BEFORE the RoIAlign / track-head launches are queued, so that the read
        # below waits for the box kernels only, not for the embeddings (which stay on the device)
        ext_h = torch.empty((len(seg_ids), 4), dtype=torch.float32, pin_memory=True)
        ext_h.copy_(ext_all[sel], non_blocking=True)
        copied = torch.cuda.Event()
        copied.record()
        embeds = self.track_head.forward_planes(T.roi_extract(fpn_feats, rois_all[sel].contiguous(), prec, self.strides))
        copied.synchronize()
        bboxes = torch.cat([ext_h, torch.tensor(score, dtype=torch.float32)[:, None]], 1)
        return seg_ids, (bboxes, torch.tensor(labels, dtype=torch.int64), embeds)            # the embeddings stay on the device

    def _sem_map_device(self, pan_dev, segments_info):
        """get_semantic_seg (:436-443) as a table look-up on the device copy of the id map; returns (sem uint8 map, the int64
        index map both look-ups share).  It needs nothing from the tracker, so callers may start its download early."""
        n = int(max([s['id'] for s in segments_info], default=0)) + 1
        sem_lut = torch.full((n,), self.num_thing_classes + self.num_stuff_classes, dtype=torch.uint8)
        for s in segments_info:
            sem_lut[s['id']] = s['category_id']
        idx = pan_dev.long()
        return sem_lut.pin_memory().to(pan_dev.device, non_blocking=True)[idx], idx

    def _trk_map_device(self, idx, segments_info, seg_ids, ids):
        """generate_track_id_maps (:445-451): float64 map of the track ids painted onto their segments"""
        n = int(max([s['id'] for s in segments_info], default=0)) + 1
        trk_lut = torch.zeros((n,), dtype=torch.float64)
        for sid, tid in zip(seg_ids, ids):
            trk_lut[sid] = float(tid)
        return trk_lut.pin_memory().to(idx.device, non_blocking=True)[idx]

    def _maps_on_device(self, pan_dev, segments_info, seg_ids, ids, to_host=True):
        """both maps as two table look-ups on the device copy of the id map (the host versions `semantic_map` / `track_id_map`
        walk 2 M pixels in numpy: 10 ms per 1024x2048 frame)"""
        sem, idx = self._sem_map_device(pan_dev, segments_info)
        trk = self._trk_map_device(idx, segments_info, seg_ids, ids)
        if not to_host:
            return sem, trk
        return sem.cpu().numpy(), trk.cpu().numpy()

    def step_device(self, fpn_feats, pan_dev, segments_info, early=None):
        """`step` on the DEVICE copy of the panoptic id map, results left on the device: (sem uint8, track float64) maps.
        Same kernels, same tracker calls, same values as `step` -- for callers that overlap the result download with the
        next frame (`VideoStreamRunner`).  `early(sem)` is called as soon as the semantic map is queued -- before the boxes,
        RoIAlign, embeddings and the tracker -- so that its download can run under them."""
        sem, idx = self._sem_map_device(pan_dev, segments_info)
        if early is not None:
            early(sem)
        seg_ids, rec = self.record(fpn_feats, None, segments_info, pan_dev)
        ids = []
        if rec is not None:
            _, _, ids = self.tracker.match(bboxes=rec[0], labels=rec[1], track_feats=rec[2], frame_id=self.cnt)
            self.cnt += 1
            ids = ids + 1
            ids[ids == -1] = 0
            ids = ids.tolist()
        return sem, self._trk_map_device(idx, segments_info, seg_ids, ids)

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
        """round 5: the heads of this per-frame call replay ONE HIP graph (a one-slot `VideoStreamRunner` kept per frame geometry,
        captured on first use from the same module calls) -- same kernels, same results, returned immediately as before; the ~70
        launches of neck -> KernelHead -> decode cost the host one graph launch instead of 2.5 ms.  Weights are re-captured when
        their versions change.  PH_VIDEO_API_EAGER=1: the eager launches of rounds 1-4."""
        if x[0].shape[0] != 1:
            raise NotImplementedError("video inference is one frame at a time (samples_per_gpu = 1, as in the reference)")
        import os
        if not records_only and x[0].is_cuda and not os.environ.get("PH_VIDEO_API_EAGER"):
            m = img_metas[0]
            key = (tuple(m["img_shape"]), tuple(m["ori_shape"]), tuple(m["batch_input_shape"]), tuple(tuple(t.shape) for t in x), x[0].dtype)
            runners = self.__dict__.setdefault("_api_runners", {})
            r = runners.get(key)
            if r is None:
                runners.clear()                      # one geometry at a time: a runner owns GBs of plan buffers
                r = runners[key] = VideoStreamRunner(self, dict(m), graph=True, pipelined=False)
            r._check_weights()
            r._start_heads(0, [x])
            r._finish(0)
            return r._collect(r._downloads.pop(0))
        _, _, (panoptic_seg, segments_info), _, depth_final = self.heads(x, img_metas, rescale)[0]
        return self.assoc.step(x, panoptic_seg, segments_info, depth_final, records_only=records_only)


def build_video_pipeline_from_config(cfg):
    """`PolyphonicVideo.__init__` (polyphonic/polyphonic_former_video.py:23-60) from a loaded reference config (nested dict with a
    `model` key, e.g. configs/polyphonic_video/poly_r50_cityscapes_1x.py after `_base_` resolution): rpn_head / roi_head with
    train_cfg / test_cfg injected as TwoStageDetector does, `track_head` from the registry, the tracker from `model.tracker`, the
    RoI extractor from `model.bbox_roi_extractor` (the shipped one: SingleRoIExtractor over RoIAlign 7x7, sampling_ratio 2 --
    csrc/ph_track.hip implements exactly that; anything else is refused).  Backbone and FPN stay the caller's.  Returns a
    `VideoFramePipeline`."""
    from .registry import build_heads_from_config, build_head, deep_cfg
    from . import track_head  # noqa: F401  (registers QuasiDenseMaskEmbedHeadGTMask and the loss names its config carries)
    model = cfg["model"]
    for k in ("track_head", "tracker", "bbox_roi_extractor"):
        if model.get(k) is None:
            raise ValueError(f"model.{k} is missing: not a PolyphonicVideo config")
    rpn_head, roi_head = build_heads_from_config(cfg)
    ext = model["bbox_roi_extractor"]
    rl = ext.get("roi_layer", {})
    if ext.get("type") != "SingleRoIExtractor" or rl.get("type") != "RoIAlign" or rl.get("output_size") not in (7, (7, 7), [7, 7]) \
            or rl.get("sampling_ratio", 0) != 2 or ext.get("out_channels") != 256:
        raise NotImplementedError(f"bbox_roi_extractor {ext!r}: libpolyhead implements SingleRoIExtractor(RoIAlign 7x7, sampling_ratio=2, 256 ch)")
    th = build_head(deep_cfg(model["track_head"]))
    return VideoFramePipeline(rpn_head, roi_head, th, deep_cfg(model["tracker"]), strides=tuple(ext["featmap_strides"]))


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
    Weights are packed at capture time; every push compares the parameters' version counters with the capture-time ones and
    captures again (all slots) when they changed (`_check_weights`, round 5)."""

    def __init__(self, pipe, img_meta, graph=True, pipelined=True, device_select=None, frames_per_launch=1):
        """`frames_per_launch` (round 6, `push` / `flush`): k > 1 buffers k pushed frames and sends them through the heads in ONE launch
        (the batch-invariant heads give every frame the bits of its own one-frame launch; fp16 / bf16 grades, see `clip_batch`), two
        launches in flight on the two slots; results come back in frame order, up to 2 k frames late"""
        import os
        self.pipe, self.metas, self.use_graph, self.pipelined = pipe, [img_meta], graph, pipelined
        self.frames_per_launch = max(1, int(frames_per_launch))
        # the merge's candidate selection + activation + argmax queued behind the decode on the heads' stream (panoptic.DeviceMerge),
        # inside the graph; False = the module API's form (class scores to the host, torch.topk there)
        self.device_select = (not os.environ.get("PH_VIDEO_HOST_SELECT")) if device_select is None else device_select
        self.reset()

    def reset(self):
        self._slots = []                 # per slot: dict(rpn, roi, g = {frames per launch: dict(x, graph, outs)}, cur, stream, done)
        self._copy_stream = None
        self._inflight = None            # frame whose heads are running: (slot index)
        self._downloads = []             # [(event, host tensors, device sources)] oldest first
        self._n = 0
        self._versions = None
        self._buf = []                   # frames_per_launch > 1: pushed frames waiting for their launch
        self._inflight_b = None          # ... and the launch whose frames are still to be finished: (slot, frames)
        self._nb = 0
        self._last_done = None
        self._rq = []                    # clips queued by records_begin: dict(chunks, next chunk to start, slots of the started ones)
        self._free = []

    def khead_timeouts(self):
        """workgroup time-outs of the one-pass KernelHead kernel over every plan this runner's slots and graphs hold (sticky counters;
        synchronises).  Non-zero only when something else holds CUs beyond the hand-off bound -- e.g. several processes sharing ONE
        GPU -- and then those calls' results came from the predicated two-pass kernels: equal to ~1e-6, not bit for bit."""
        seen, n = set(), 0
        for sl in self._slots:
            plans = list(getattr(sl["rpn"], "_plans", {}).values())
            for st in sl["g"].values():
                for d in st.get("plans") or []:
                    plans += list(d.values())
            for p in plans:
                if id(p) not in seen and hasattr(p, "timeouts"):
                    seen.add(id(p))
                    n += int(p.timeouts())
        return n

    def _weight_versions(self):
        from . import _lib
        return _lib.param_versions(self.pipe.rpn_head) + _lib.param_versions(self.pipe.roi_head)

    def _check_weights(self):
        """the graphs replay weight PACKS made at capture time and the second slot is a deep copy of the head
        modules -- after load_state_dict / an optimizer step / a replaced sub-module they would silently keep the old weights.
        The parameters' version counters are compared with the capture-time ones; on a change the frame in flight is finished
        (with the weights it started with), the slots are dropped and the next frame captures again from the current modules."""
        v = self._weight_versions()
        if self._versions is None:
            self._versions = v
        elif v != self._versions:
            if self._inflight is not None:
                self._finish(self._inflight)
                self._inflight = None
            if getattr(self, "_inflight_b", None) is not None:
                i, n = self._inflight_b
                for b in range(n):
                    self._finish(i, b)
                self._inflight_b = None
            self._slots = []
            self._versions = v

    # -- slots ---------------------------------------------------------------------------------------------------
    def _slot(self, i):
        while len(self._slots) <= i:
            if not self._slots:
                rpn, roi = self.pipe.rpn_head, self.pipe.roi_head
            else:
                rpn, roi = self._clone_heads()
            # the slots' streams carry the heads' graphs (milliseconds of HBM-bound launches); the merges / records of the frames in
            # front of them are a dozen small kernels on the caller's stream with the host waiting for each: PH_VIDEO_SLOT_PRIO=low
            # puts the heads one priority level below them (A/B: see DESIGN 7b "Round 6")
            prio = 0
            if os.environ.get("PH_VIDEO_SLOT_PRIO") == "low":
                try:
                    lo, hi = torch.cuda.Stream.priority_range()
                    prio = max(lo, hi)
                except Exception:
                    prio = 0
            self._slots.append(dict(rpn=rpn, roi=roi, g={}, cur=None, stream=torch.cuda.Stream(priority=prio), done=None))
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
         semantic_aspp_out) = sl["rpn"].simple_test_rpn(x, self.metas * x[0].shape[0])
        o = sl["roi"]._decode(x_feats, proposal_feats, mask_preds, depth_feats, depth_proposal)
        depth_init = E.upsample2x(depth_pred.float().contiguous())                         # kernel_update.py:302-307
        return o["cls"], o["mask_up"], o["depth_up"], depth_init

    def _start_heads(self, i, frames, borrowed=False):
        """copy the frames (a list of one-frame FPN level tuples: ONE for the per-frame calls, a clip's chunk for `records`) into
        slot i's static inputs for that many frames per launch and start its heads on the slot's stream.

        `borrowed` (round 6, clips): the caller keeps the frames' tensors unchanged until their records have been returned, so nothing
        is copied -- the neck's ingest (fp32 NCHW -> 16-bit conv input planes, its first kernel) runs per frame straight from the caller's
        tensors, on the caller's stream and OUTSIDE the graph, which then starts at the towers' first convs; RoIAlign reads the caller's
        levels.  The staging copy of a 1024 x 2048 frame's four levels is 178 MB read + 178 MB written: 0.9 ms of an 8-frame clip's
        8.4 ms.  Needs the tower-stream neck plan (clip launches of 2+ frames) and fp32 contiguous levels; otherwise the copy form runs."""
        sl = self._slot(i)
        main = torch.cuda.current_stream()
        B = len(frames)
        neck0 = getattr(sl["rpn"], "localization_fpn", None)
        pre = bool(borrowed and self.use_graph and B >= 2 and neck0 is not None and hasattr(neck0, "ingest_frames")
                   and os.environ.get("PH_VIDEO_BORROW", "1") != "0"
                   and all(t.dtype == torch.float32 and t.is_contiguous() for f in frames for t in f[:4]))
        if sl.get("no_borrow"):
            pre = False                                      # (this neck plan cannot take frames one by one: found at a first capture)
        key = ("borrowed", B) if pre else B
        sl["cur"] = key
        from . import panoptic as Pn
        if not self.use_graph:
            # slot-owned copies in this path too (torch.cat copies; one frame is cloned): RoIAlign reads the levels one push
            # later, and the contract is that the caller may reuse its tensors once push() returns
            x = tuple(t.clone() for t in frames[0]) if B == 1 else tuple(torch.cat([f[l] for f in frames], 0) for l in range(len(frames[0])))
            outs = self._heads_device(sl, x)
            dm = (sl["g"].get(key) or {}).get("dm")
            if self.device_select:
                dm = dm or Pn.DeviceMerge(sl["roi"], *outs, self.metas[0])
                dm.begin(*outs)
                dm.download()
            sl["g"][key] = dict(x=x, graph=None, outs=outs, dm=dm, B=B, pre=False)
            sl["done"] = torch.cuda.Event()
            sl["done"].record(main)
            return
        st = sl["g"].get(key)
        if st is None:
            st = sl["g"][key] = dict(x=tuple(t.new_empty((B,) + tuple(t.shape[1:])) for t in frames[0]), graph=None, outs=None, dm=None, B=B,
                                     pre=pre)
            for b, f in enumerate(frames):
                for d, t in zip(st["x"], f):
                    d[b:b + 1].copy_(t)
            # a clip's 2-3 frames per launch (`records`): the neck's four level towers run on their own streams inside the graph (the
            # small levels' convs are 16-64 workgroups; eagerly the forks cost more host time than the overlap returns, in a graph
            # nothing: cfg4 +2-6 %, same bits).  Not at one frame per launch: the per-frame loops got SLOWER with it (1.68 -> 1.92 ms
            # per frame pipelined: the previous frame's small association kernels wait behind four streams of neck kernels)
            neck = getattr(sl["rpn"], "localization_fpn", None)
            # 0: never, 1 (default): clip launches only, 2: + the sequential one-frame loop (module API: heads 1.75 -> 1.64 ms, but merge
            # and association + 0.04 each behind it: - 0.05 ms of 2.87 per frame, inside the scatter; left off)
            _ct = os.environ.get("PH_VIDEO_CLIP_TOWERS", "1")
            towers = neck is not None and _ct != "0" and (B >= 2 or (_ct == "2" and not self.pipelined))
            if towers:
                neck._clip_towers = True
            try:
                outs = self._heads_device(sl, st["x"])      # warm-up outside the capture: plans, packs, kernel attributes
                if self.device_select:
                    st["dm"] = Pn.DeviceMerge(sl["roi"], *outs, self.metas[0])
                    st["dm"].begin(*outs)
                torch.cuda.synchronize()
                nplan = None
                if pre:
                    # borrowed clips: the graph starts BEHIND the neck's ingest (engine.NeckPlan.skip_ingest); when this plan cannot be
                    # filled frame by frame (shared level buffers, two-plane grade) the slot falls back to the copy form for good
                    nplan = neck.clip_plan(B, tuple(tuple(t.shape[-2:]) for t in st["x"][:4]), st["x"][0].device)
                    if nplan is None or not nplan.can_ingest_frames():
                        sl["no_borrow"] = True               # this slot copies from now on; the launch starts over in the copy form
                        del sl["g"][key]
                        return self._start_heads(i, frames, borrowed=False)
                g = torch.cuda.CUDAGraph()
                try:
                    if nplan is not None:
                        nplan.skip_ingest = True
                    with torch.cuda.graph(g):
                        st["outs"] = self._heads_device(sl, st["x"])
                        if st["dm"] is not None:
                            st["dm"].begin(*st["outs"])
                finally:
                    if nplan is not None:
                        nplan.skip_ingest = False
            finally:
                if towers:
                    neck._clip_towers = False
            st["graph"] = g
            st["nplan"] = nplan
            # the graph replays the plans' device buffers: KernelHead / KernelUpdateIterHead keep ONE plan and drop it when the
            # batch size changes (a clip's last chunk), so the graph holds its own references -- without them a later replay wrote
            # into freed blocks (harmless while the allocator kept them mapped; a memory fault once it had not: clips of 3 + 3 + 2)
            neck = getattr(sl["rpn"], "localization_fpn", None)
            st["plans"] = [dict(m._plans) for m in (sl["rpn"], sl["roi"], neck) if m is not None and hasattr(m, "_plans")]
        for f in frames:
            if len(f) != len(st["x"]) or any(tuple(d.shape[1:]) != tuple(t.shape[1:]) or d.dtype != t.dtype for d, t in zip(st["x"], f)):
                raise ValueError("VideoStreamRunner: the FPN levels changed shape / dtype; one runner serves one stream of equally "
                                 "sized frames (call reset() to re-capture)")
        if st.get("pre"):
            st["frames"] = list(frames)                      # RoIAlign reads these; the references keep the storage alive
            neck0.ingest_frames(frames, plan=st["nplan"])    # on the caller's stream, in front of `ready`; the plan the graph replays
        else:
            st["frames"] = None
            for b, f in enumerate(frames):
                for d, t in zip(st["x"], f):
                    d[b:b + 1].copy_(t, non_blocking=True)   # on the caller's stream: the frame may be reused once push returns
        ready = torch.cuda.Event()
        ready.record(main)
        with torch.cuda.stream(sl["stream"]):
            sl["stream"].wait_event(ready)
            # heads of different launches run ONE AFTER THE OTHER (round 6): what a launch should overlap with is the host's walk
            # through the frames in front of it and their small kernels, not another heads graph -- two persistent one-pass KernelHead
            # launches at once starve each other until one gives up (8-frame clips queued back to back: 18.5 ms per step instead of 8)
            last = getattr(self, "_last_done", None)
            if last is not None:
                sl["stream"].wait_event(last)
            st["graph"].replay()
            if st["dm"] is not None:
                st["dm"].download()
            sl["done"] = torch.cuda.Event()
            sl["done"].record(sl["stream"])
            self._last_done = sl["done"]

    def _frame_levels(self, i, b=0):
        """the FPN levels of frame b of slot i's current launch (views of its static inputs)"""
        sl = self._slots[i]
        st = sl["g"][sl["cur"]]
        if st.get("frames"):                                 # a borrowed clip: the caller's tensors
            return tuple(st["frames"][b])
        x = st["x"]
        return x if sl["cur"] == 1 else tuple(t[b:b + 1] for t in x)

    # -- the part of a frame that follows the heads -----------------------------------------------------------------
    def _merge(self, i, b=0):
        from . import panoptic as Pn
        sl = self._slots[i]
        torch.cuda.current_stream().wait_event(sl["done"])
        st = sl["g"][sl["cur"]]
        if st["dm"] is not None:
            sl["done"].synchronize()                         # candidates + histograms are in pinned memory
            return st["dm"].finish(b)
        cls, mask_up, depth_up, depth_init = st["outs"]
        return Pn.get_panoptic_device(sl["roi"], cls[b], mask_up[b], depth_up[b], depth_init[b], self.metas[0])

    def _finish(self, i, b=0):
        """merge -> association -> start the download of the result maps of frame b of slot i's launch"""
        pan_dev, info, _, d_final = self._merge(i, b)
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream()
        main, cs = torch.cuda.current_stream(), self._copy_stream
        host, kept = [], []

        def download(t):
            # fresh pinned buffers (the caller owns them); the device sources stay referenced until the copy has finished.  Each
            # map starts its way to the host as soon as it is queued: the depth map right after the merge, the semantic map before
            # the association, only the track-id map (16 of the 27 MB of a 1024 x 2048 frame) after the tracker
            ready = torch.cuda.Event()
            ready.record(main)
            h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            with torch.cuda.stream(cs):
                cs.wait_event(ready)
                t.record_stream(cs)
                h.copy_(t, non_blocking=True)
            host.append(h)
            kept.append(t)

        download(d_final)
        sem, trk = self.pipe.assoc.step_device(self._frame_levels(i, b), pan_dev, info, early=download)
        download(trk)
        ev = torch.cuda.Event()
        ev.record(cs)
        self._downloads.append((ev, [host[1], host[2], host[0]], tuple(kept)))       # (sem, track, depth)

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
        if self.frames_per_launch > 1 and self._launch_size() > 1:
            return self._push_batched(x)
        self._check_weights()
        if not self.pipelined:
            self._start_heads(0, [x])
            self._finish(0)
            self._n += 1
            return self._collect(self._downloads.pop(0)) if len(self._downloads) > 1 else None
        i = self._n & 1
        self._start_heads(i, [x])                            # frame t: heads on slot i's stream ...
        if self._inflight is not None:
            self._finish(self._inflight)                     # ... while the host takes frame t - 1 through merge / association
        self._inflight = i
        self._n += 1
        return self._collect(self._downloads.pop(0)) if len(self._downloads) > 1 else None

    def _launch_size(self):
        """frames per launch of the batched `push`: `frames_per_launch` where the heads are batch invariant (`clip_batch`'s conditions), else 1"""
        from . import _lib, engine as E
        grade = E.KHEAD_PREC.get(getattr(self.pipe.rpn_head, "precision", None))
        ok = (grade in (_lib.PH_PREC_BF16, _lib.PH_PREC_F16) and not os.environ.get("PH_KHEAD_TWOPASS")
              and getattr(self.pipe.rpn_head, "frame_invariant", False) and getattr(self.pipe.roi_head, "frame_invariant", False))
        return self.frames_per_launch if ok else 1

    def _launch_buffered(self):
        """the buffered frames' heads on the next slot; then the previous launch's frames walk merge -> association underneath them"""
        if not self._inflight_b and not self._buf:
            return
        if self._buf:
            self._check_weights()                        # (new weights: finishes the launch in flight with the old ones, drops the slots)
        prev = self._inflight_b
        self._inflight_b = None
        if self._buf:
            i = (self._nb & 1) if self.pipelined else 0
            if not self.pipelined and prev is not None:
                for b in range(prev[1]):
                    self._finish(prev[0], b)
                prev = None
            # the buffered frames are this runner's own clones (`_push_batched`): they ARE the slot-owned copy -- borrowed, so the
            # launch does not copy them a second time into the graph's static inputs (round 6)
            self._start_heads(i, self._buf, borrowed=True)
            self._inflight_b = (i, len(self._buf))
            self._buf = []
            self._nb += 1
        if prev is not None:
            for b in range(prev[1]):
                self._finish(prev[0], b)

    def _push_batched(self, x):
        assert self._inflight is None and not self._rq, "push() in batches and push_record() / records() must not be interleaved"
        self._buf.append(tuple(t.clone() for t in x))    # the caller may reuse its tensors once push returns
        if len(self._buf) >= self.frames_per_launch:
            self._launch_buffered()
        self._n += 1
        return self._collect(self._downloads.pop(0)) if len(self._downloads) > self.frames_per_launch else None

    def flush(self):
        """the results still in flight, oldest first (a list of result lists)"""
        if self._buf or self._inflight_b:
            self._launch_buffered()                      # the partial last batch (its own launch size), then whatever is still in flight
            self._launch_buffered()
        if self._inflight is not None:
            self._finish(self._inflight)
            self._inflight = None
        out = [self._collect(p) for p in self._downloads]
        self._downloads = []
        return out

    def clip_batch(self, frames):
        """frames per launch for `records`: the heads are frame independent, so a clip's frames go through neck ->
        KernelHead -> decode TOGETHER -- at one frame per launch those kernels are latency bound and eight frames cost barely more
        than two.  Every frame's tensors must stay those of the per-frame loop bit for bit.  Round 6: that holds by construction for
        any batch -- every choice that touches a frame's arithmetic follows the FRAME's geometry, never B (the neck's conv tile rows
        and output-stage tile runs, csrc/ph_neck.hip conv_th / ph_khead.hip kh_tiles_per_wg_plain; the pooling's pixel split and the
        final-stage form of `frame_invariant` plans, engine.DecodePlan / KernelHeadPlan; the one-pass KernelHead groups a frame's
        GroupNorm sums by its own pixel slices) -- so the frames per launch are a pure launch-size choice (`PH_VIDEO_CLIP_BATCH=n`; 1 restores
        one frame per launch; default below), asserted for 1 .. 16 frames by tests/test_gpu_video.py::test_heads_are_batch_invariant.  Exceptions:
        the grades whose KernelHead runs the two-pass kernel (fp32 / mixed: its workgroups' tile runs are sized by the batch, which
        regroups the fp32 partial sums -- 1e-6 differences, not bit identity) and heads switched to `frame_invariant = False`."""
        import os
        from . import _lib, engine as E
        # default: a clip of 4 or more frames goes as TWO launches (half the clip each, at most 8 frames): the second half's heads run
        # while the host walks the first half through merge -> boxes -> RoIAlign -> track head, and with clips queued (`records_begin`
        # ahead of the previous `records_end`) the next clip's first half follows on the slot that frees up.  Measured on one box
        # (profiles/r06/cfg4_sweep.txt, 8-frame clips): 2 / 3 / 4 / 8 frames per launch 785 / 830 / 943 / 734 frames/s; 16-frame clips:
        # 768 / 786 / 889 / 979.  Shorter clips: one launch.
        n = len(frames)
        cap = int(os.environ.get("PH_VIDEO_CLIP_BATCH", "0")) or (n if n <= 3 else min(8, (n + 1) // 2))
        grade = E.KHEAD_PREC.get(getattr(self.pipe.rpn_head, "precision", None))
        if grade not in (_lib.PH_PREC_BF16, _lib.PH_PREC_F16) or os.environ.get("PH_KHEAD_TWOPASS"):
            return 1
        if not (getattr(self.pipe.rpn_head, "frame_invariant", False) and getattr(self.pipe.roi_head, "frame_invariant", False)):
            return 1
        return max(1, min(cap, len(frames)))

    def records(self, frames, borrowed=True):
        """the sharded mode's per-step work for a rank's clip: `simple_test(..., records_only=True)` of every frame in order.
        The clip goes through the heads in chunks of `clip_batch` frames per launch; the heads of up to two chunks are in flight
        at a time: chunk k + 1's are started BEFORE chunk k's merges / records, and the clip's first two chunks start back to back.
        Returns [(segment ids, (bboxes, labels, embeds) or None)] per frame.  = `records_begin` + `records_end`.  The call returns when
        every frame has been consumed, so the frames are `borrowed` (no staging copy, `_start_heads`) unless the caller says otherwise."""
        self.records_begin(frames, borrowed=borrowed)
        return self.records_end()

    def _pump(self):
        """start the heads of waiting chunks, oldest clip first, while a slot is free (graph replays on the slots' streams)"""
        for clip in self._rq:
            while clip["next"] < len(clip["chunks"]) and self._free:
                i = self._free.pop(0)
                self._start_heads(i, clip["chunks"][clip["next"]], borrowed=clip.get("borrowed", False))
                clip["slots"].append(i)
                clip["next"] += 1
            if clip["next"] < len(clip["chunks"]):
                break                                    # launch order = frame order: a later clip never overtakes an earlier one

    def records_begin(self, frames, borrowed=False):
        """first half of `records`: queues the clip and STARTS the heads of as many of its chunks as slots are free -- returns without
        waiting for the device.  Round 6: clips QUEUE -- `records_begin` of the next clip may be called before `records_end` of the
        previous one, and then its heads run (on the other slot) underneath the previous clip's merges / records, which are host work
        and a few small kernels: the sharded video loop (bench.cfg4_run) does exactly that, so a step costs max(heads, merges + records +
        all-gather + replay) instead of their sum.  `records_end` always returns the OLDEST queued clip's records.
        `borrowed=True`: the caller leaves the frames' tensors unchanged until `records_end` has returned this clip (`_start_heads`)."""
        frames = list(frames)
        assert self._inflight is None, "records() and push() / push_record() must not be interleaved"
        for f in frames:
            self._check(f)
        if not getattr(self, "_rq", None):
            self._rq = []
            self._check_weights()                        # (drops the slots when the weights changed: only between clips)
            self._free = list(range(2 if self.pipelined else 1))
        Bc = self.clip_batch(frames) if frames else 1
        self._rq.append(dict(chunks=[frames[k:k + Bc] for k in range(0, len(frames), Bc)], next=0, slots=[], borrowed=bool(borrowed)))
        self._pump()

    def records_end(self):
        """second half of `records`, for the oldest queued clip: merges / records chunk by chunk (the synchronising part); every slot it
        frees goes to the next waiting chunk at once -- of this clip or of the clip queued behind it"""
        clip = self._rq[0]
        out = []
        for c, chunk in enumerate(clip["chunks"]):
            if c >= len(clip["slots"]):
                self._pump()                             # (a slot is free: every earlier chunk of this clip has been consumed)
            i = clip["slots"][c]
            for b in range(len(chunk)):
                out.append(self._record(i, b))
            self._free.append(i)
            self._pump()
        self._rq.pop(0)
        return out

    def push_record(self, x):
        """the sharded mode's per-frame work (`simple_test(..., records_only=True)`) for the frame pushed ONE call ago (None for
        the first call): its heads ran while the caller dealt with the frame before; `flush_record()` returns the last frame's.
        Returns (segment ids, (bboxes, labels, embeds) or None); nothing map-sized is downloaded."""
        self._check(x)
        i = self._n & 1 if self.pipelined else 0
        prev = None
        if not self.pipelined:
            self._start_heads(0, [x])
            self._n += 1
            return self._record(0)
        self._start_heads(i, [x])
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

    def _record(self, i, b=0):
        pan_dev, info, _, _ = self._merge(i, b)
        return self.pipe.assoc.record(self._frame_levels(i, b), None, info, pan_dev)
