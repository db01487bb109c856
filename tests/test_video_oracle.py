"""CPU: the video-association oracle (oracle/video_oracle.py) against the reference's own helpers / track head
(tests/golden/video.npz) and, for RoIAlign (mmcv op, not runnable here), analytic properties."""
import numpy as np
import pytest
import torch

import helpers as Hh
from oracle import video_oracle as VO

torch.set_grad_enabled(False)


def test_boxes_and_embed_head_match_reference():
    z = Hh.load_golden("video.npz")
    pan, info, feats, roi_feats = Hh.video_case()
    masks = torch.stack([torch.from_numpy(pan == s["id"]) for s in info])
    stat = VO.mask_stat_boxes(masks)
    assert np.allclose(stat.numpy(), z["stat_boxes"], rtol=0, atol=1e-4)
    rois = torch.cat([torch.zeros(len(stat), 1), stat], 1).clamp(min=0.0)
    assert np.allclose(rois.numpy(), z["rois"], atol=1e-4)
    assert np.array_equal(VO.mask_extent_boxes(masks).numpy(), z["extent_boxes"])
    sd = Hh.seeded_fill(Hh.TRACK_HEAD_SHAPES, 4321)
    emb = VO.track_embed_head(sd, roi_feats)
    assert Hh.rel_err(emb, z["embeds"]) < 1e-5


def test_roi_align_analytic_properties():
    # constant field -> constant; affine field f = a*x + b*y + c -> value at the bin centre (interior RoIs)
    H, W = 40, 64
    ys, xs = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing="ij")
    const = torch.full((1, 2, H, W), 3.5)
    aff = (0.25 * xs + 0.5 * ys + 1.0)[None, None].repeat(1, 2, 1, 1)
    rois = torch.tensor([[0, 40.0, 24.0, 120.0, 88.0], [0, 16.5, 8.25, 77.0, 40.0]])
    s = 0.25
    out = VO.roi_align(const, rois, s)
    assert torch.allclose(out, torch.full_like(out, 3.5))
    out = VO.roi_align(aff, rois, s)
    for i, r in enumerate(rois):
        x1, y1, x2, y2 = [float(v) * s - 0.5 for v in r[1:]]
        bw, bh = (x2 - x1) / 7, (y2 - y1) / 7
        for ph in range(7):
            for pw in range(7):
                cx, cy = x1 + (pw + 0.5) * bw, y1 + (ph + 0.5) * bh
                assert abs(float(out[i, 0, ph, pw]) - (0.25 * cx + 0.5 * cy + 1.0)) < 1e-4
    lv = VO.map_roi_levels(torch.tensor([[0, 0, 0, 50., 50.], [0, 0, 0, 120., 120.], [0, 0, 0, 300., 300.], [0, 0, 0, 2000., 900.]]))
    assert lv.tolist() == [0, 1, 2, 3]


def test_roi_align_against_atens_bilinear_sampler():
    """RoIAlign (aligned, 7x7 bins, 2x2 samples per bin) is bilinear sampling at known continuous coordinates followed by
    a 2x2 mean.  `torch.nn.functional.grid_sample(align_corners=False)` is ATen's own bilinear sampler with the same
    pixel-centre convention (sample coordinate s in RoIAlign's frame = pixel-centre coordinate s + 0.5): an implementation
    this build did not write.  For boxes whose samples lie inside [0, H-1] x [0, W-1] (where the two border rules agree)
    the oracle must reproduce it; random features, so a wrong tap or weight cannot cancel."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(8)
    H, W, C = 48, 80, 5
    feat = torch.randn(1, C, H, W, generator=g)
    s = 0.125
    rois = torch.tensor([[0, 40.0, 24.0, 300.0, 200.0], [0, 100.5, 60.25, 190.0, 140.75], [0, 33.0, 17.0, 47.5, 39.0],
                         [0, 320.0, 100.0, 600.0, 360.0]])
    out = VO.roi_align(feat, rois, s)
    for i, r in enumerate(rois):
        x1, y1, x2, y2 = [float(v) * s - 0.5 for v in r[1:]]
        bw, bh = (x2 - x1) / 7, (y2 - y1) / 7
        ys = torch.tensor([y1 + ph * bh + (iy + 0.5) * bh / 2 for ph in range(7) for iy in range(2)])     # 14 sample rows
        xs = torch.tensor([x1 + pw * bw + (ix + 0.5) * bw / 2 for pw in range(7) for ix in range(2)])
        assert ys.min() >= 0 and ys.max() <= H - 1 and xs.min() >= 0 and xs.max() <= W - 1
        # grid_sample: normalised coordinate of continuous position p (pixel centres at integers) = (2p + 1) / size - 1
        gy, gx = (2 * ys + 1) / H - 1, (2 * xs + 1) / W - 1
        grid = torch.stack(torch.meshgrid(gy, gx, indexing="ij")[::-1], -1)[None]                         # [1,14,14,(x,y)]
        samp = F.grid_sample(feat, grid, mode="bilinear", padding_mode="zeros", align_corners=False)      # [1,C,14,14]
        want = F.avg_pool2d(samp, 2)[0]
        assert torch.allclose(out[i], want, atol=2e-5), (i, float((out[i] - want).abs().max()))


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_tracker_oracle_reproduces_the_reference_ids(seed):
    """oracle/video_oracle.TrackerOracle (the checker of the end-to-end association tests in tests/test_gpu_video.py) against the ids,
    labels and boxes the REFERENCE QuasiDenseEmbedTracker produced on the same synthetic clips (tests/golden/tracker.npz)"""
    import json
    z = Hh.load_golden("tracker.npz")
    tr = VO.TrackerOracle(**json.loads(bytes(z["cfg_json"]).decode()))
    cnt = 1
    for f, bb, lab, emb in Hh.tracker_records(seed):
        if bb.shape[0] == 0:
            continue
        obb, olab, ids = tr.match(bb, lab, emb, cnt)
        cnt += 1
        ids = ids + 1
        ids[ids == -1] = 0
        assert np.array_equal(ids.numpy(), z[f"s{seed}_f{f}_ids"]), f
        assert np.array_equal(olab.numpy(), z[f"s{seed}_f{f}_labels"])
        assert np.array_equal(obb.numpy(), z[f"s{seed}_f{f}_bboxes"])
