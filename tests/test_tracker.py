"""CPU: the tracker drop-in (host logic of the video association step) reproduces the integer
track ids of the REFERENCE's QuasiDenseEmbedTracker bit-for-bit on synthetic clips (golden: tests/golden/tracker.npz,
written by oracle/gen_golden.py from the reference class itself), also when the frames come back from an all-gather in
arbitrary order."""
import json

import numpy as np
import pytest
import torch

import helpers as Hh
from polyphonicformer_amd import video as V
from polyphonicformer_amd import dist as D


def _cfg(z):
    return json.loads(bytes(z["cfg_json"]).decode())


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_tracker_ids_match_reference(seed):
    z = Hh.load_golden("tracker.npz")
    tr = V.TRACKERS.build(dict(type="QuasiDenseEmbedTracker", **_cfg(z)))
    cnt = 1
    for f, bb, lab, emb in Hh.tracker_records(seed):
        if bb.shape[0] == 0:
            continue
        obb, olab, ids = tr.match(bboxes=bb, labels=lab, track_feats=emb, frame_id=cnt)
        cnt += 1
        ids = ids + 1
        ids[ids == -1] = 0
        assert np.array_equal(ids.numpy(), z[f"s{seed}_f{f}_ids"]), f
        assert np.array_equal(olab.numpy(), z[f"s{seed}_f{f}_labels"])
        assert np.array_equal(obb.numpy(), z[f"s{seed}_f{f}_bboxes"])


def test_replay_is_order_independent_and_matches_reference():
    z = Hh.load_golden("tracker.npz")
    recs = Hh.tracker_records(2)
    shuffled = [recs[i] for i in torch.randperm(len(recs), generator=torch.Generator().manual_seed(0)).tolist()]
    out = V.replay_tracking(shuffled, _cfg(z))
    for f, bb, lab, emb in recs:
        assert np.array_equal(out[f].numpy(), z[f"s2_f{f}_ids"])


def test_pack_unpack_records_are_lossless():
    for f, bb, lab, emb in Hh.tracker_records(3)[:4]:
        rec, n = D.pack_track_records(bb, lab, emb)
        b2, l2, e2 = D.unpack_track_records(rec, n)
        assert torch.equal(b2, bb) and torch.equal(l2, lab) and torch.equal(e2, emb)


def test_bbox_overlaps_edge_cases():
    a = torch.tensor([[0., 0., 10., 10.], [5., 5., 5., 5.]])
    iou = V.bbox_overlaps(a, a)
    assert iou[0, 0] == 1.0 and iou[1, 1] == 0.0 and iou[0, 1] == 0.0
    assert V.bbox_overlaps(a[:0], a).shape == (0, 2)


def test_sem_track_maps_and_wire_format():
    """get_semantic_seg / generate_track_id_maps (polyphonic_former_video.py:436-451) and the pre_eval record
    (datasets/cityscapes_dvps.py:325-338) against their direct per-segment formulation"""
    pan, info, _, _ = Hh.video_case(seed=5)
    info[1] = dict(id=info[1]["id"], isthing=False, category_id=12, area=3)         # a stuff segment
    seg_ids, idxs, labels, score = V.things_for_tracking(pan, info)
    assert info[1]["id"] not in seg_ids and len(seg_ids) == len(info) - 1
    ids = list(range(5, 5 + len(seg_ids)))
    ids[2] = 0
    sem_ref = np.ones(pan.shape, dtype=np.uint8) * 8 + 11
    for s in info:
        sem_ref[pan == s["id"]] = s["category_id"]
    trk_ref = np.zeros(pan.shape)
    for sid, t in zip(seg_ids, ids):
        trk_ref[pan == sid] = t
    sem, trk = V.semantic_map(pan, info, 8, 11), V.track_id_map(pan, seg_ids, ids)
    assert sem.dtype == np.uint8 and np.array_equal(sem, sem_ref) and np.array_equal(trk, trk_ref)
    rec = V.wire_record({"sem": sem, "track": trk, "depth": np.ones(pan.shape, dtype=np.float64)})
    assert rec["panseg"].dtype == np.uint32 and rec["depth"].dtype == np.float32
    assert np.array_equal(rec["panseg"], sem_ref.astype(np.int64) * 10000 + trk_ref.astype(np.int64))
