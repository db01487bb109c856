"""the DVPS wire format and the DVPQ metric, against goldens produced by the unmodified
reference evaluator reading files written by this build's writer (oracle/gen_golden_eval.py)."""
import json
import os

import numpy as np
import torch

import helpers as Hh
from polyphonicformer_amd import dvps_eval as D
from polyphonicformer_amd import video as V


def _golden():
    return json.load(open(os.path.join(Hh.GOLDEN, "dvps_eval.json")))


def test_wire_record_format(tmp_path):
    """keys, dtypes, file names and the sem * 10000 + track packing of datasets/cityscapes_dvps.py:325-338"""
    g = _golden()
    fr = Hh.dvps_clip(seed=21)[0]
    path = D.save_record(str(tmp_path), fr["seq"], fr["img"], fr["pred"])
    assert os.path.basename(path) == "%06d_%06d.pth" % (fr["seq"], fr["img"]) and os.path.dirname(path).endswith("pred")
    rec = torch.load(path, weights_only=False)
    assert sorted(rec.keys()) == g["record_keys"]
    assert {k: str(v.dtype) for k, v in rec.items()} == g["record_dtypes"]
    assert np.array_equal(rec["panseg"] // 10000, fr["pred"]["sem"]) and np.array_equal(rec["panseg"] % 10000, fr["pred"]["track"])
    w = V.wire_record(fr["pred"])                    # the alias PolyphonicVideo's host mirror exports
    assert np.array_equal(w["panseg"], rec["panseg"]) and w["depth"].dtype == np.float32


def test_vpq_eval_matches_the_reference_bit_for_bit():
    gold = Hh.load_golden("dvps_vpq.npz")
    frames = Hh.dvps_clip(seed=21)
    for i, fr in enumerate(frames[:6]):
        p, g = D.wire_record(fr["pred"])["panseg"], D.wire_record(fr["gt"])["panseg"]
        got = np.stack(D.vpq_eval(p, g, num_classes=19))
        assert np.array_equal(got, gold[f"frame{i}"]), i       # counts AND the float64 IoU sums


def test_vpq_eval_edge_cases():
    z = np.zeros((4, 4), dtype=np.uint32)
    iou, tp, fn, fp = D.vpq_eval(z + 3 * 10000, z + 3 * 10000, num_classes=19)      # one perfect stuff segment
    assert tp[3] == 1 and iou[3] == 1.0 and fn.sum() == 0 and fp.sum() == 0
    iou, tp, fn, fp = D.vpq_eval(z + 3 * 10000, z + 255 * 10000, num_classes=19)    # prediction entirely on ignore
    assert tp.sum() == 0 and fn.sum() == 0 and fp.sum() == 0
    iou, tp, fn, fp = D.vpq_eval(z + 3 * 10000 + 1, z + 4 * 10000 + 1, num_classes=19)   # wrong class
    assert fn[4] == 1 and fp[3] == 1 and tp.sum() == 0


def test_video_evaluate_matches_the_reference_printout(tmp_path):
    g = _golden()
    for fr in Hh.dvps_clip(seed=21):
        D.save_record(str(tmp_path), fr["seq"], fr["img"], fr["pred"], "pred")
        D.save_record(str(tmp_path), fr["seq"], fr["img"], fr["gt"], "gt")
    res = D.video_evaluate(str(tmp_path), num_classes=19, num_things=8)
    assert len(res) == len(g["dvpq"]) == 16
    for (k, thr), vals in res.items():
        key = f"{k}:{'inf' if thr == 0 else thr}"
        want = g["dvpq"][key]                      # the reference prints 3 decimals
        assert all(abs(round(a, 3) - b) <= 1e-3 for a, b in zip(vals, want)), (key, vals, want)
    assert any(v[0] > 1.0 for v in res.values())


def test_depth_errors():
    g = _golden()["depth_errors"]
    frames = Hh.dvps_clip(seed=21)
    gt = np.concatenate([f["gt"]["depth"].ravel() for f in frames])
    pr = np.concatenate([f["pred"]["depth"].ravel() for f in frames])
    got = D.compute_errors(pr, gt)
    for k, v in g.items():
        assert abs(float(got[k]) - v) <= 1e-6 * max(1.0, abs(v)), k
