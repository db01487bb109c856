"""Panoptic merge -- KernelUpdateIterHead.get_panoptic / merge_stuff_thing_stuff_joint
(polyphonic/kernel_update.py:421-535) on libpolyhead's merge kernels.

Host logic here (segment selection, the accept loop over <= max_per_img + num_stuff segments) mirrors
the reference's Python; the per-pixel work (activation, two-step bilinear rescale, first-index
argmax, area histograms, id / depth paste) is one activate + one argmax + one paste kernel and a
single 2K-int D2H copy instead of ~3 syncs per segment."""
import ctypes as C

import numpy as np
import torch

from . import _lib

DEPTH_MODES = {"sigmoid": 0, "monodepth": 1}


def select_segments(cls_scores, num_proposals, num_thing_classes, max_per_img):
    """kernel_update.py:428-434 (top-k thing (query, class) pairs) and :448-459 (stuff diagonal,
    sorted by score).  cls_scores: CPU fp32 [N, L] (post-sigmoid).  Returns query index, label, score
    of the K candidate segments, things first (the order `total_*` are concatenated in, :487-489)."""
    thing = cls_scores[:num_proposals][:, :num_thing_classes]
    tscore, tidx = thing.flatten(0, 1).topk(max_per_img, sorted=True)
    tq = torch.div(tidx, num_thing_classes, rounding_mode="floor")
    tl = tidx % num_thing_classes
    sscore = cls_scores[num_proposals:][:, num_thing_classes:].diag()
    sscore, sind = torch.sort(sscore, descending=True)
    return (torch.cat([tq, sind + num_proposals]), torch.cat([tl, sind + num_thing_classes]),
            torch.cat([tscore, sscore]))


def accept_loop(scores, labels, area, orig, num_thing_classes, instance_score_thr, overlap_thr):
    """kernel_update.py:497-533 on the two per-segment pixel histograms.  Returns (newid[K], segments_info).
    Whether a segment is kept depends on its own counts only (:503-512); the loop of the reference merely numbers the
    kept ones in descending-score order (:497), so the tests are evaluated for all K at once and only the kept
    segments are visited."""
    K = len(scores)
    newid = np.zeros(K, dtype=np.int32)
    # :497; the reference's argsort is not stable, so its order among EQUAL scores is unspecified: stable = ascending
    # index among ties, deterministic (and what torch's CPU sort gives for the reference's call too)
    order = torch.argsort(-scores, stable=True).numpy()
    sc, lab = scores.numpy(), labels.numpy()
    area, orig = np.asarray(area, dtype=np.int64), np.asarray(orig, dtype=np.int64)
    isthing = lab < num_thing_classes
    # :503 is `total_scores[k] < merge_cfg.instance_score_thr`: a 0-dim fp32 TENSOR against a Python scalar, which torch
    # evaluates in fp32 (the scalar is rounded to the tensor's dtype) -- a thing scoring exactly fp32(0.7) is NOT below 0.7
    keep = ~(isthing & (sc.astype(np.float32) < np.float32(instance_score_thr)))  # :503
    keep &= (area > 0) & (orig > 0)                                          # :510
    with np.errstate(divide="ignore", invalid="ignore"):
        keep &= ~((area / np.where(orig > 0, orig, 1)) < overlap_thr)        # :511 (python float division, as there)
    info = []
    for seg, k in enumerate((k for k in order if keep[k]), start=1):
        k = int(k)
        cls = int(lab[k])
        newid[k] = seg
        if cls < num_thing_classes:
            info.append({'id': seg, 'isthing': True, 'score': float(scores[k]), 'category_id': cls, 'instance_id': k})
        else:
            info.append({'id': seg, 'isthing': False, 'category_id': cls, 'area': int(area[k])})
    return newid, info


def _geom(src_hw, img_meta):
    h, w = img_meta['img_shape'][:2]
    Hb, Wb = img_meta['batch_input_shape']
    Ho, Wo = img_meta['ori_shape'][:2]
    return (C.c_int32 * 8)(src_hw[0], src_hw[1], Hb, Wb, h, w, Ho, Wo), (Ho, Wo)


def merge_on_device(act_mask, act_depth, act_depth0, scores, labels, geom, out_hw, num_thing_classes,
                    instance_score_thr, overlap_thr, from_probs=False):
    """argmax -> accept loop -> paste on already activated maps (device fp32).  Returns DEVICE tensors
    (panoptic ids int32 [Ho, Wo], depth_basic, depth_final fp32) and the host-side segments_info: the one
    synchronisation in here is the 2K-int histogram copy the accept loop needs."""
    lib, dev = _lib.load(), act_mask.device
    K = act_mask.shape[0]
    Ho, Wo = out_hw
    ids = torch.empty((Ho, Wo), dtype=torch.int32, device=dev)
    counts = torch.empty((2, K), dtype=torch.int32, device=dev)
    sc = scores.to(torch.float32).contiguous()
    sc = sc if sc.is_cuda else sc.pin_memory().to(dev, non_blocking=True)    # (a pageable H2D copy synchronises)
    _lib.check(lib.ph_panoptic_argmax(_lib.ptr(act_mask), _lib.ptr(sc), K, geom, 1 if from_probs else 0, _lib.ptr(ids),
                                      _lib.ptr(counts), _lib.stream_ptr()), "ph_panoptic_argmax")
    cnt_h = torch.empty((2, K), dtype=torch.int32, pin_memory=True)
    cnt_h.copy_(counts, non_blocking=True)
    torch.cuda.current_stream().synchronize()                                # the one mid-merge sync (2K ints)
    cnt = cnt_h.numpy()
    newid, info = accept_loop(scores, labels, cnt[0], cnt[1], num_thing_classes, instance_score_thr, overlap_thr)
    nid = torch.from_numpy(newid).pin_memory().to(dev, non_blocking=True)
    pan = torch.empty((Ho, Wo), dtype=torch.int32, device=dev)
    d_basic = torch.empty((Ho, Wo), dtype=torch.float32, device=dev)
    d_final = torch.empty((Ho, Wo), dtype=torch.float32, device=dev)
    _lib.check(lib.ph_panoptic_paste(_lib.ptr(ids), _lib.ptr(nid), _lib.ptr(act_depth), _lib.ptr(act_depth0), geom,
                                     1 if from_probs else 0, _lib.ptr(pan), _lib.ptr(d_basic), _lib.ptr(d_final),
                                     _lib.stream_ptr()), "ph_panoptic_paste")
    return pan, info, d_basic, d_final


def merge_device(act_mask, act_depth, act_depth0, scores, labels, geom, out_hw, num_thing_classes,
                 instance_score_thr, overlap_thr, from_probs=False):
    """`merge_on_device` + the reference API's host results (numpy): the three maps go to pinned host memory (torch's
    caching host allocator) with async copies and ONE final sync"""
    pan, info, d_basic, d_final = merge_on_device(act_mask, act_depth, act_depth0, scores, labels, geom, out_hw,
                                                  num_thing_classes, instance_score_thr, overlap_thr, from_probs)
    outs = []
    for t in (pan, d_basic, d_final):
        h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        h.copy_(t, non_blocking=True)
        outs.append(h)
    torch.cuda.current_stream().synchronize()
    return outs[0].numpy(), info, outs[1].numpy(), outs[2].numpy()


def _activated(head, cls_scores, mask_preds, depth_preds, depth_init, img_meta):
    """segment selection (host) + ph_panoptic_activate: everything `merge_*` needs"""
    if not head.merge_joint:
        raise NotImplementedError               # as the reference (:467-468)
    lib, dev = _lib.load(), mask_preds.device
    cfg = head.test_cfg
    cls_h = torch.empty(cls_scores.shape, dtype=torch.float32, pin_memory=True)
    cls_h.copy_(cls_scores.detach().float(), non_blocking=True)
    torch.cuda.current_stream().synchronize()
    q, labels, scores = select_segments(cls_h, head.num_proposals, head.num_thing_classes, cfg.max_per_img)
    K = len(q)
    N, h2, w2 = mask_preds.shape
    mask_preds, depth_preds = mask_preds.contiguous(), depth_preds.contiguous()
    codes = {torch.float32: _lib.PH_OUT_F32, torch.bfloat16: _lib.PH_OUT_BF16, torch.float16: _lib.PH_OUT_F16}
    if mask_preds.dtype != depth_preds.dtype or mask_preds.dtype not in codes:
        raise _lib.PolyheadError("mask/depth logits must both be fp32, both bf16 or both fp16")
    dt = codes[mask_preds.dtype]
    d0 = depth_init.reshape(h2, w2).float().contiguous()
    qd = q.to(torch.int32).pin_memory().to(dev, non_blocking=True)
    act_mask = torch.empty((K, h2, w2), dtype=torch.float32, device=dev)
    act_depth = torch.empty((K, h2, w2), dtype=torch.float32, device=dev)
    act_d0 = torch.empty((h2, w2), dtype=torch.float32, device=dev)
    mode = DEPTH_MODES[head.mask_head[-1].depth_act_mode]
    _lib.check(lib.ph_panoptic_activate(_lib.ptr(mask_preds), _lib.ptr(depth_preds), dt, _lib.ptr(d0), _lib.ptr(qd), K, h2, w2,
                                        mode, _lib.ptr(act_mask), _lib.ptr(act_depth), _lib.ptr(act_d0), _lib.stream_ptr()),
               "ph_panoptic_activate")
    geom, out_hw = _geom((h2, w2), img_meta)
    merge_cfg = cfg.merge_stuff_thing
    return (act_mask, act_depth, act_d0, scores, labels, geom, out_hw, head.num_thing_classes, merge_cfg.instance_score_thr,
            merge_cfg.overlap_thr)


def get_panoptic(head, cls_scores, mask_preds, depth_preds, depth_init, img_meta):
    """kernel_update.py:421-469 for one image.  cls_scores [N, L] (post-sigmoid), mask_preds /
    depth_preds [N, 2H, 2W] logits (fp32 or bf16), depth_init [1, 2H, 2W] fp32 logits; all on the GPU.
    Returns (None, None, (panoptic_seg int32 ndarray, segments_info), depth_basic, depth_final)."""
    pan, info, d_basic, d_final = merge_device(*_activated(head, cls_scores, mask_preds, depth_preds, depth_init, img_meta))
    return None, None, (pan, info), d_basic, d_final


def get_panoptic_device(head, cls_scores, mask_preds, depth_preds, depth_init, img_meta):
    """the same with the three result maps left on the DEVICE (int32 ids, fp32 depth_basic, depth_final) for callers that
    go on working there -- the video association step needs the id map on the GPU, not on the host"""
    return merge_on_device(*_activated(head, cls_scores, mask_preds, depth_preds, depth_init, img_meta))


class DeviceMerge:
    """The merge's device half for B frames of one head with NO host step up to the two pixel histograms: candidate selection
    (`ph_panoptic_select` instead of a D2H of the class scores + torch.topk on the host), activation and argmax are queued behind
    the decode -- `begin` is pure kernel launches on static buffers, so `video.VideoStreamRunner` captures it into the heads' HIP
    graph and the merge's first 0.25 ms of device work run on the heads' stream, under the previous frame's host path.  One small
    D2H (`download`: candidates + histograms, 5 K ints per frame) feeds `finish` = accept loop (host, as in the reference) + paste.
    Same kernels and the same values as `get_panoptic_device`; the candidate ORDER among exactly equal scores is by ascending index
    here and whatever torch.topk / torch.sort give there (unspecified in the reference too)."""

    def __init__(self, head, cls_scores, mask_preds, depth_preds, depth_init, img_meta):
        if not head.merge_joint:
            raise NotImplementedError               # as the reference (:467-468)
        self.head, dev = head, mask_preds.device
        B, N, L = cls_scores.shape
        _, _, h2, w2 = mask_preds.shape
        cfg = head.test_cfg
        self.B, self.N, self.L, self.h2, self.w2 = B, N, L, h2, w2
        self.K = K = cfg.max_per_img + min(N - head.num_proposals, L - head.num_thing_classes)
        codes = {torch.float32: _lib.PH_OUT_F32, torch.bfloat16: _lib.PH_OUT_BF16, torch.float16: _lib.PH_OUT_F16}
        if mask_preds.dtype != depth_preds.dtype or mask_preds.dtype not in codes:
            raise _lib.PolyheadError("mask/depth logits must both be fp32, both bf16 or both fp16")
        self.dt = codes[mask_preds.dtype]
        self.mode = DEPTH_MODES[head.mask_head[-1].depth_act_mode]
        self.geom, self.out_hw = _geom((h2, w2), img_meta)
        Ho, Wo = self.out_hw
        e = lambda shape, dt: torch.empty(shape, dtype=dt, device=dev)
        self.pack = e((B, 5 * K), torch.int32)              # per frame: q | labels | scores (fp32 bits) | counts [2][K]
        self.host = torch.empty((B, 5 * K), dtype=torch.int32, pin_memory=True)
        self.act_mask, self.act_depth = e((B, K, h2, w2), torch.float32), e((B, K, h2, w2), torch.float32)
        self.act_d0, self.ids = e((B, h2, w2), torch.float32), e((B, Ho, Wo), torch.int32)
        self._keep = None

    def begin(self, cls_scores, mask_preds, depth_preds, depth_init):
        """cls_scores [B,N,L] (post-sigmoid), mask_preds / depth_preds [B,N,2H,2W] logits, depth_init [B,1,2H,2W]: launches only"""
        lib, K, s = _lib.load(), self.K, _lib.stream_ptr
        cls = cls_scores.detach().float().contiguous()
        m, d, d0 = mask_preds.contiguous(), depth_preds.contiguous(), depth_init.float().contiguous()
        self._keep = (cls, m, d, d0)                        # alive until the launches have run (static under graph capture)
        i32 = lambda t, off: C.c_void_p(t.data_ptr() + 4 * off)
        _lib.check(lib.ph_panoptic_select(_lib.ptr(cls), self.N * self.L, self.B, self.N, self.L, self.head.num_proposals,
                                          self.head.num_thing_classes, self.head.test_cfg.max_per_img, i32(self.pack, 0), i32(self.pack, K),
                                          i32(self.pack, 2 * K), 5 * K, s()), "ph_panoptic_select")
        esz = m.element_size()
        plane = self.N * self.h2 * self.w2
        for b in range(self.B):
            row = b * 5 * K
            _lib.check(lib.ph_panoptic_activate(C.c_void_p(m.data_ptr() + b * plane * esz), C.c_void_p(d.data_ptr() + b * plane * esz), self.dt,
                                                _lib.ptr(d0[b]), i32(self.pack, row), K, self.h2, self.w2, self.mode, _lib.ptr(self.act_mask[b]),
                                                _lib.ptr(self.act_depth[b]), _lib.ptr(self.act_d0[b]), s()), "ph_panoptic_activate")
            _lib.check(lib.ph_panoptic_argmax(_lib.ptr(self.act_mask[b]), i32(self.pack, row + 2 * K), K, self.geom, 0, _lib.ptr(self.ids[b]),
                                              i32(self.pack, row + 3 * K), s()), "ph_panoptic_argmax")

    def download(self):
        """candidates + histograms to pinned host memory, asynchronously on the current stream (not part of a captured graph)"""
        self.host.copy_(self.pack, non_blocking=True)

    def finish(self, b=0):
        """frame b, after the caller has synchronised with `download`: accept loop -> paste.  Returns what `merge_on_device` returns"""
        lib, K, dev = _lib.load(), self.K, self.pack.device
        h = self.host[b].numpy()
        scores = torch.from_numpy(h[2 * K:3 * K].view(np.float32).copy())
        labels = torch.from_numpy(h[K:2 * K].astype(np.int64))
        cnt = h[3 * K:].reshape(2, K)
        merge_cfg = self.head.test_cfg.merge_stuff_thing
        newid, info = accept_loop(scores, labels, cnt[0], cnt[1], self.head.num_thing_classes, merge_cfg.instance_score_thr,
                                  merge_cfg.overlap_thr)
        nid = torch.from_numpy(newid).pin_memory().to(dev, non_blocking=True)
        Ho, Wo = self.out_hw
        pan = torch.empty((Ho, Wo), dtype=torch.int32, device=dev)
        d_basic = torch.empty((Ho, Wo), dtype=torch.float32, device=dev)
        d_final = torch.empty((Ho, Wo), dtype=torch.float32, device=dev)
        _lib.check(lib.ph_panoptic_paste(_lib.ptr(self.ids[b]), _lib.ptr(nid), _lib.ptr(self.act_depth[b]), _lib.ptr(self.act_d0[b]), self.geom,
                                         0, _lib.ptr(pan), _lib.ptr(d_basic), _lib.ptr(d_final), _lib.stream_ptr()), "ph_panoptic_paste")
        return pan, info, d_basic, d_final

