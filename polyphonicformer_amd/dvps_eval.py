"""SURVEY.md 8(f) row N2 -- the output / evaluation wire format of the video path, host side.

* `save_record` / `record_name`: what `CityscapesDVPSDataset.pre_eval` writes per frame
  (datasets/cityscapes_dvps.py:325-338): `{"panseg": uint32 sem * 10000 + track_id, "depth": float32}` as
  `{seq:06d}_{img:06d}.pth`, so the unmodified reference evaluator (polyphonic/apis/video_evaluate.py) can score
  this build's outputs.
* `vpq_eval`, `evaluate_clip`, `video_evaluate`: the DVPQ metric itself (datasets/utils.py:31-106,
  polyphonic/apis/video_evaluate.py:14-111), restated on sorted-unique integer arrays instead of Python dicts
  (same pairing, same accumulation order: the per-class sums are bit-identical to the reference's) -- used by the
  tests to check the writer against goldens produced by the reference evaluator, and usable on its own.
* `compute_errors`: the depth error metrics (datasets/utils.py:109-137).
Pure numpy / torch.save: nothing here touches the GPU."""
import os

import numpy as np
import torch

INSTANCE_DIVISOR = 10000          # datasets/utils.py:5
_EPSILON = 1e-15                  # video_evaluate.py:11


def record_name(seq_id, img_id):
    return "{:06d}_{:06d}.pth".format(int(seq_id), int(img_id))


def wire_record(result):
    """result: dict(sem=int map, track=int map, depth=float map) as PolyphonicVideo.simple_test returns them
    (polyphonic_former_video.py:397-405)"""
    pan = result["sem"].astype(np.int64) * INSTANCE_DIVISOR + result["track"].astype(np.int64)
    return {"panseg": pan.astype(np.uint32), "depth": result["depth"].astype(np.float32)}


def save_record(save_dir, seq_id, img_id, result, sub="pred"):
    d = os.path.join(save_dir, sub)
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, record_name(seq_id, img_id))
    torch.save(wire_record(result), path)
    return path


def vpq_eval(pred_ids, gt_ids, num_classes=19, max_ins=INSTANCE_DIVISOR, ign_id=255):
    """-> (iou_per_class, tp_per_class, fn_per_class, fp_per_class), float64 [num_classes + 1]"""
    num_cat = num_classes + 1
    iou_c, tp_c, fn_c, fp_c = (np.zeros(num_cat, dtype=np.float64) for _ in range(4))
    pred = np.asarray(pred_ids).astype(np.int64).ravel()
    gt = np.asarray(gt_ids).astype(np.int64).ravel()
    pu, pinv, parea = np.unique(pred, return_inverse=True, return_counts=True)
    gu, ginv, garea = np.unique(gt, return_inverse=True, return_counts=True)
    # intersections, sorted by (gt id, pred id) = the reference's key order gt * 1e9 + pred
    iu, iarea = np.unique(ginv.astype(np.int64) * len(pu) + pinv, return_counts=True)
    gi, pi = iu // len(pu), iu % len(pu)
    gcat, pcat = gu // max_ins, pu // max_ins
    is_void = gu == ign_id * max_ins
    is_ign = gcat == ign_id
    void_ov = np.zeros(len(pu), dtype=np.int64)
    np.add.at(void_ov, pi[is_void[gi]], iarea[is_void[gi]])
    ign_ov = np.zeros(len(pu), dtype=np.int64)
    np.add.at(ign_ov, pi[is_ign[gi]], iarea[is_ign[gi]])
    same = gcat[gi] == pcat[pi]
    union = garea[gi] + parea[pi] - iarea - void_ov[pi]
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = iarea / union
    tp = same & (iou > 0.5)
    g_matched = np.zeros(len(gu), dtype=bool)
    p_matched = np.zeros(len(pu), dtype=bool)
    for k in np.nonzero(tp)[0]:                   # ascending key order, like the reference's dict iteration
        c = int(gcat[gi[k]])
        tp_c[c] += 1
        iou_c[c] += iou[k]
        g_matched[gi[k]] = True
        p_matched[pi[k]] = True
    for k in np.nonzero(~g_matched & ~is_ign)[0]:
        fn_c[int(gcat[k])] += 1
    for k in np.nonzero(~p_matched)[0]:
        if ign_ov[k] / parea[k] > 0.5:
            continue
        fp_c[int(pcat[k])] += 1
    return iou_c, tp_c, fn_c, fp_c


def evaluate_clip(pred_records, gt_records, depth_thr, num_classes):
    """video_evaluate.py:14-37: frames of a clip side by side (axis 1); pixels whose relative depth error exceeds
    `depth_thr` are relabelled to class `num_classes` before the panoptic comparison"""
    pred_pan = np.concatenate([r["panseg"] for r in pred_records], axis=1)
    gt_pan = np.concatenate([r["panseg"] for r in gt_records], axis=1)
    pred_dep = np.concatenate([r["depth"] for r in pred_records], axis=1)
    gt_dep = np.concatenate([r["depth"] for r in gt_records], axis=1)
    if depth_thr > 0.:
        m = gt_dep > 0.
        sel = pred_pan[m]
        bad = (np.abs(pred_dep[m] - gt_dep[m]) / gt_dep[m]) > depth_thr
        sel[bad] = num_classes * INSTANCE_DIVISOR
        pred_pan[m] = sel
    return vpq_eval(pred_pan, gt_pan, num_classes=num_classes)


def _pth_names(d):
    return sorted(f for f in os.listdir(d) if ".pth" in f and not f.startswith("._"))


def video_evaluate(eval_dir, num_classes, num_things, windows=(1, 2, 3, 4), depth_thrs=(0, 0.5, 0.25, 0.1)):
    """video_evaluate.py:40-111 -> {(k, lambda): (DVPQ, DVPQ_thing, DVPQ_stuff)} in percent; clips never span
    two sequences"""
    gt_dir, pred_dir = os.path.join(eval_dir, "gt"), os.path.join(eval_dir, "pred")
    gts = [os.path.join(gt_dir, f) for f in _pth_names(gt_dir)]
    preds = [os.path.join(pred_dir, f) for f in _pth_names(pred_dir)]
    cache = {}

    def load(p):
        if p not in cache:
            cache[p] = torch.load(p, weights_only=False)
        return cache[p]

    out = {}
    n = len(preds)
    for k in windows:
        for thr in depth_thrs:
            res = []
            for idx in range(n):
                if idx + k - 1 >= n:
                    break
                s0 = int(os.path.basename(preds[idx]).split("_")[0])
                s1 = int(os.path.basename(preds[idx + k - 1]).split("_")[0])
                if s0 != s1:
                    continue
                pr = [{kk: np.array(v) for kk, v in load(preds[idx + j]).items()} for j in range(k)]
                gr = [load(gts[idx + j]) for j in range(k)]
                res.append(evaluate_clip(pr, gr, thr, num_classes))
            if not res:
                continue
            iou, tp, fn, fp = (np.stack([r[j] for r in res]).sum(axis=0)[:num_classes] for j in range(4))
            sq = iou / (tp + _EPSILON)
            rq = tp / (tp + 0.5 * fn + 0.5 * fp + _EPSILON)
            pq = np.nan_to_num(sq * rq)
            out[(k, thr)] = (float(pq.mean() * 100), float(pq[:num_things].mean() * 100), float(pq[num_things:].mean() * 100))
    return out


def compute_errors(pred, gt):
    """datasets/utils.py:109-137"""
    pred, gt = pred[gt > 0.], gt[gt > 0.]
    thresh = np.maximum(gt / pred, pred / gt)
    return dict(abs_rel=np.mean(np.abs(gt - pred) / gt), sq_rel=np.mean(((gt - pred) ** 2) / gt),
                rmse=np.sqrt(((gt - pred) ** 2).mean()), rmse_log=np.sqrt(((np.log(gt) - np.log(pred)) ** 2).mean()),
                a1=(thresh < 1.25).mean(), a2=(thresh < 1.25 ** 2).mean(), a3=(thresh < 1.25 ** 3).mean())
