"""GPU: device side of the video association step vs the reference's goldens (tests/golden/video.npz)
and the oracle: segment boxes, FPN RoIAlign, track embedding head; tolerance 1e-3 (fp32 precision) / 3e-2 (bf16)."""
import numpy as np
import pytest
import torch

import helpers as Hh
from oracle import video_oracle as VO
from polyphonicformer_amd import _lib, engine as E, track_head as T
from polyphonicformer_amd.registry import HEADS

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def test_segment_boxes_golden(gpu):
    z = Hh.load_golden("video.npz")
    pan, info, feats, roi_feats = Hh.video_case()
    rois, ext = T.segment_boxes(torch.from_numpy(pan).to(gpu), len(info))
    assert np.allclose(rois.cpu().numpy(), z["rois"], atol=1e-3)
    assert np.array_equal(ext.cpu().numpy(), z["extent_boxes"])
    # an id that does not occur -> the reference's empty-mask conventions
    rois2, ext2 = T.segment_boxes(torch.from_numpy(pan).to(gpu), len(info) + 2)
    assert rois2[-1].abs().sum() == 0 and ext2[-1].tolist() == [-1.0, -1.0, 10.0, 10.0]


@pytest.mark.parametrize("prec", [_lib.PH_PREC_SPLIT, _lib.PH_PREC_BF16])
def test_roi_align_fpn_vs_oracle(gpu, prec):
    # a 960 x 1600 image: boxes from 30 px to 900 px so that all four FPN levels are used, some touching the borders
    rois = torch.tensor([[0, 10.0, 20.0, 45.0, 60.0], [0, 300.5, 100.25, 420.0, 260.0], [0, 0.0, 0.0, 250.0, 240.0],
                         [0, 700.0, 300.0, 1200.0, 800.0], [0, 100.0, 50.0, 1590.0, 950.0], [0, 1500.0, 900.0, 1599.0, 959.0],
                         [0, 640.0, 480.0, 641.0, 481.5]])
    feats = [torch.randn(1, 256, 960 // s, 1600 // s, generator=torch.Generator().manual_seed(s)) for s in (4, 8, 16, 32)]
    ref = VO.roi_extract(feats, rois)
    assert set(VO.map_roi_levels(rois).tolist()) == {0, 1, 2, 3}
    planes, f32 = T.roi_extract([f.to(gpu) for f in feats], rois.to(gpu), prec, want_f32=True)
    assert Hh.rel_err(f32.cpu(), ref) < 1e-4          # fp32 coordinate arithmetic vs the oracle's python doubles
    rec = sum(planes[p].view(torch.bfloat16).float() for p in range(planes.shape[0])).cpu()      # [n,49,256]
    tol = 2e-4 if prec == _lib.PH_PREC_SPLIT else 5e-3
    assert Hh.rel_err(rec.permute(0, 2, 1).reshape(-1, 256, 7, 7), ref) < tol


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_track_embed_head_golden(gpu, precision):
    z = Hh.load_golden("video.npz")
    _, info, _, roi_feats = Hh.video_case()
    head = HEADS.build(dict(type="QuasiDenseMaskEmbedHeadGTMask", num_convs=4, num_fcs=1, embed_channels=256,
                            norm_cfg=dict(type="GN", num_groups=32),
                            loss_track=dict(type="MultiPosCrossEntropyLoss", loss_weight=0.25),
                            loss_track_aux=dict(type="L2Loss", neg_pos_ub=3)))
    sd = Hh.seeded_fill(Hh.TRACK_HEAD_SHAPES, 4321)
    assert {"track_head." + k: tuple(v.shape) for k, v in head.state_dict().items()} == Hh.TRACK_HEAD_SHAPES
    head.load_state_dict({k[len("track_head."):]: v for k, v in sd.items()})
    head.to(gpu).eval()
    head.precision = precision
    emb = head(roi_feats.to(gpu))
    e = Hh.rel_err(emb.cpu(), z["embeds"])
    print("track embed head rel err", precision, e)
    assert e < (1e-3 if precision == "fp32" else 3e-2)


def test_association_chain_vs_oracle(gpu):
    """pan map -> boxes -> FPN RoIAlign -> embed head -> tracker, as polyphonic_former_video.py:359-396 wires it"""
    from polyphonicformer_amd import video as V
    pan, info, feats, _ = Hh.video_case(seed=12, H=192, W=320, nseg=12)
    sd = Hh.seeded_fill(Hh.TRACK_HEAD_SHAPES, 4321)
    masks = torch.stack([torch.from_numpy(pan == s["id"]) for s in info])
    rois_ref = torch.cat([torch.zeros(len(info), 1), VO.mask_stat_boxes(masks)], 1).clamp(min=0)
    emb_ref = VO.track_embed_head(sd, VO.roi_extract(feats, rois_ref))
    head = HEADS.build(dict(type="QuasiDenseMaskEmbedHeadGTMask", norm_cfg=dict(type="GN", num_groups=32)))
    head.load_state_dict({k[len("track_head."):]: v for k, v in sd.items()})
    head.to(gpu).eval()
    rois, ext = T.segment_boxes(torch.from_numpy(pan).to(gpu), len(info))
    emb = head.forward_planes(T.roi_extract([f.to(gpu) for f in feats], rois, E.PREC["fp32"]))
    assert Hh.rel_err(emb.cpu(), emb_ref) < 1e-3
    bb = torch.cat([ext.cpu(), torch.tensor([[s["score"]] for s in info])], 1)
    lab = torch.tensor([s["category_id"] for s in info])
    a = V.QuasiDenseEmbedTracker(init_score_thr=0.35, obj_score_thr=0.3)
    b = V.QuasiDenseEmbedTracker(init_score_thr=0.35, obj_score_thr=0.3)
    for f in (1, 2):
        ia = a.match(bb, lab, emb.cpu(), f)[2]
        ib = b.match(torch.cat([VO.mask_extent_boxes(masks), bb[:, 4:]], 1), lab, emb_ref, f)[2]
        assert torch.equal(ia, ib)


def test_two_frame_clip_association(gpu):
    """BASELINE config 3: a 2-frame clip through the association step (VideoAssociator.step) -- the second frame is the
    first one shifted by a few pixels, so every object must keep its track id; checked against the oracle chain."""
    from polyphonicformer_amd import video as V
    sd = Hh.seeded_fill(Hh.TRACK_HEAD_SHAPES, 4321)
    head = HEADS.build(dict(type="QuasiDenseMaskEmbedHeadGTMask", norm_cfg=dict(type="GN", num_groups=32)))
    head.load_state_dict({k[len("track_head."):]: v for k, v in sd.items()})
    head.to(gpu).eval()
    cfg = dict(init_score_thr=0.35, obj_score_thr=0.3, match_score_thr=0.5, memo_tracklet_frames=5, memo_backdrop_frames=1,
               memo_momentum=0.8, nms_conf_thr=0.5, nms_backdrop_iou_thr=0.3, nms_class_iou_thr=0.7, with_cats=True,
               match_metric="bisoftmax")
    assoc = V.VideoAssociator(head, cfg, 8, 11)
    pan0, info, feats, _ = Hh.video_case(seed=21, H=192, W=320, nseg=8)
    pan1 = np.roll(pan0, (2, 3), axis=(0, 1))
    feats1 = [torch.roll(f, (1, 1), dims=(2, 3)) if i == 0 else f for i, f in enumerate(feats)]
    ref_tr = V.QuasiDenseEmbedTracker(**cfg)
    outs, ref_ids = [], []
    for fi, (pan, ff) in enumerate(((pan0, feats), (pan1, feats1))):
        depth = np.full(pan.shape, 1.5, dtype=np.float32)
        outs.append(assoc.step([f.to(gpu) for f in ff], pan, info, depth)[0])
        masks = torch.stack([torch.from_numpy(pan == s["id"]) for s in info])
        rois = torch.cat([torch.zeros(len(info), 1), VO.mask_stat_boxes(masks)], 1).clamp(min=0)
        emb = VO.track_embed_head(sd, VO.roi_extract(ff, rois))
        bb = torch.cat([VO.mask_extent_boxes(masks), torch.tensor([[s["score"]] for s in info])], 1)
        ids = ref_tr.match(bb, torch.tensor([s["category_id"] for s in info]), emb, fi + 1)[2] + 1
        ids[ids == -1] = 0
        ref_ids.append(ids.tolist())
    for fi, pan in enumerate((pan0, pan1)):
        want = V.track_id_map(pan, [s["id"] for s in info], ref_ids[fi])
        assert np.array_equal(outs[fi]["track"], want)
        assert outs[fi]["sem"].dtype == np.uint8 and outs[fi]["depth"].dtype == np.float32
    assert V.wire_record(outs[1])["panseg"].dtype == np.uint32
