"""The boundary pinned to the reference's OWN config files: `tests/golden/ref_model_cfg.json` holds the `model`
dicts of configs/polyphonic_image/poly_r50_cityscapes_2x.py and configs/polyphonic_video/poly_r50_cityscapes_1x.py after `_base_`
resolution (dumped by oracle/gen_ref_cfg.py from /root/reference -- data, no source text).  Every head of this package is built
from those kwargs the way TwoStageDetector.__init__ (mmdet/models/detectors/two_stage.py:36-49) and PolyphonicVideo.__init__
(polyphonic/polyphonic_former_video.py:49-60) build the reference's; the CPU tests check the parameter names / shapes against the
reference's state dict and the attributes the detectors read, the GPU test runs `simple_test` on the device."""
import copy
import json
import os

import pytest
import torch

import helpers as Hh


def ref_cfg(which):
    with open(os.path.join(Hh.GOLDEN, "ref_model_cfg.json")) as f:
        return {"model": copy.deepcopy(json.load(f)[which])}


def test_golden_is_the_reference_configs_model_dict():
    """when /root/reference is present (this container), the committed JSON is what the generator writes today"""
    import sys
    ref = os.environ.get("POLY_REFERENCE_ROOT", "/root/reference")
    if not os.path.isdir(os.path.join(ref, "configs")):
        pytest.skip("the reference is not on this box")
    sys.path.insert(0, os.path.join(Hh.REPO, "oracle"))
    import gen_ref_cfg as G
    with open(os.path.join(Hh.GOLDEN, "ref_model_cfg.json")) as f:
        committed = json.load(f)
    for which, rel in committed["_source"].items():
        model = G.load_cfg(os.path.join(ref, rel))["model"]
        assert G.jsonable({k: v for k, v in model.items() if k not in ("backbone", "neck")}) == committed[which], which


@pytest.mark.parametrize("which", ["image", "video"])
def test_heads_build_from_the_reference_config(which):
    from polyphonicformer_amd.registry import build_heads_from_config
    import polyphonicformer_amd.kernel_head, polyphonicformer_amd.kernel_update, polyphonicformer_amd.semantic_fpn  # noqa: F401,E401
    cfg = ref_cfg(which)
    m = cfg["model"]
    kh, ih = build_heads_from_config(cfg)
    # parameter names and shapes: the reference's own state dict (tests/golden/full_state_keys.json was written from the reference
    # classes built with this very configuration, oracle/gen_golden.py) -- 263 head entries + the neck under localization_fpn
    with open(os.path.join(Hh.GOLDEN, "full_state_keys.json")) as f:
        ref = json.load(f)
    with open(os.path.join(Hh.GOLDEN, "neck_state_keys.json")) as f:
        nref = json.load(f)["full"]
    got = {"roi_head." + k: list(v.shape) for k, v in ih.state_dict().items()}
    got.update({"rpn_head." + k: list(v.shape) for k, v in kh.state_dict().items() if not k.startswith("localization_fpn.")})
    assert got == ref and len(ref) == 263
    assert {k[len("localization_fpn."):]: list(v.shape) for k, v in kh.state_dict().items() if k.startswith("localization_fpn.")} == nref
    # what the detectors and the test-time code read off the heads
    assert kh.num_proposals == m["rpn_head"]["num_proposals"] == 100 and kh.num_classes == 19
    assert kh.num_thing_classes == m["num_thing_classes"] and kh.num_stuff_classes == m["num_stuff_classes"]
    assert kh.cat_stuff_mask and kh.localization_fpn.__class__.__name__ == "SemanticFPNWrapper"
    assert ih.num_stages == 3 and len(ih.mask_head) == 3 and ih.num_proposals == 100
    assert ih.mask_head[0].mask_upsample_stride == 2 and ih.mask_head[0].num_classes == 19
    assert ih.mask_head[0].loss_cls.use_sigmoid and kh.loss_seg.use_sigmoid and not ih.mask_head[0].loss_rank.use_sigmoid
    assert ih.test_cfg.max_per_img == 100 and ih.test_cfg.merge_stuff_thing.overlap_thr == 0.6
    assert ih.test_cfg.merge_stuff_thing.instance_score_thr == 0.3 and ih.test_cfg.mask_thr == 0.5
    # train_cfg injected as TwoStageDetector does: both heads own an assigner (the reference builds it in init_assigner_sampler)
    # (the iter head replicates its dict per stage, kernel_update.py:92-101)
    assert kh.train_cfg["assigner"]["type"] == "MaskHungarianAssignerWithDepth" and kh.train_cfg.pos_weight == 1.0
    assert len(ih.train_cfg) == 3 and all(c["assigner"]["type"] == "MaskHungarianAssignerWithDepth" for c in ih.train_cfg)
    assert kh.assigner.__class__.__name__ == "MaskHungarianAssignerWithDepth" and kh.sampler.__class__.__name__ == "MaskPseudoSampler"
    assert len(ih.mask_assigner) == 3 and all(a.__class__.__name__ == "MaskHungarianAssignerWithDepth" for a in ih.mask_assigner)
    if which == "video":
        assert m["roi_head"]["tracking"] is True and m["rpn_head"]["loss_depth"]["loss_weight"] == 1.0


def test_video_pipeline_builds_from_the_reference_config():
    from polyphonicformer_amd import video as V
    cfg = ref_cfg("video")
    pipe = V.build_video_pipeline_from_config(cfg)
    th, trk = pipe.assoc.track_head, pipe.assoc.tracker
    assert th.__class__.__name__ == "QuasiDenseMaskEmbedHeadGTMask" and th.num_convs == 4 and th.embed_channels == 256
    keys = set(th.state_dict())
    assert {"convs.0.conv.weight", "convs.3.gn.bias", "fcs.0.weight", "fc_embed.bias"} <= keys and len(keys) == 4 * 3 + 4
    t = cfg["model"]["tracker"]
    assert trk.__class__.__name__ == t["type"] == "QuasiDenseEmbedTracker"
    for k in ("init_score_thr", "obj_score_thr", "match_score_thr", "memo_tracklet_frames", "memo_backdrop_frames", "memo_momentum",
              "nms_conf_thr", "nms_backdrop_iou_thr", "nms_class_iou_thr", "with_cats", "match_metric"):
        assert getattr(trk, k) == t[k], k
    assert tuple(pipe.assoc.strides) == (4, 8, 16, 32)
    bad = ref_cfg("video")
    bad["model"]["bbox_roi_extractor"]["roi_layer"]["output_size"] = 14
    with pytest.raises(NotImplementedError):
        V.build_video_pipeline_from_config(bad)
    with pytest.raises(ValueError):
        V.build_video_pipeline_from_config(ref_cfg("image"))


@pytest.mark.gpu
def test_simple_test_of_the_pipeline_built_from_the_reference_config(gpu):
    """one `PolyphonicVideo.simple_test` (after extract_feat) on the device with every head built from the reference's video config:
    result maps of the reference's types and shapes, ids consistent over two frames"""
    import numpy as np
    from polyphonicformer_amd import video as V
    torch.manual_seed(5)
    pipe = V.build_video_pipeline_from_config(ref_cfg("video"))
    for mod in (pipe.rpn_head, pipe.roi_head, pipe.assoc.track_head):
        mod.init_weights()
        mod.eval().to(gpu)
    with torch.no_grad():       # un-trained weights: let segments through (as bench._video_pipeline does)
        pipe.roi_head.mask_head[-1].fc_cls.bias.fill_(1.0)
    pipe.roi_head.test_cfg.merge_stuff_thing.overlap_thr = 0.0
    H8, W8 = 256, 512
    g = torch.Generator().manual_seed(3)
    x = [torch.randn(1, 256, H8 // s, W8 // s, generator=g).to(gpu) for s in (4, 8, 16, 32)]
    meta = [dict(img_shape=(H8, W8, 3), ori_shape=(H8, W8, 3), batch_input_shape=(H8, W8))]
    pipe.init_tracker()
    with torch.no_grad():
        r0 = pipe.simple_test(x, meta)[0]
        r1 = pipe.simple_test(x, meta)[0]
    for r in (r0, r1):
        assert r["sem"].shape == (H8, W8) and r["sem"].dtype == np.uint8
        assert r["track"].shape == (H8, W8) and r["track"].dtype == np.float64
        assert r["depth"].shape == (H8, W8) and r["depth"].dtype == np.float32 and np.isfinite(r["depth"]).all()
    assert (r0["sem"] == r1["sem"]).all()                        # the same frame twice: same segments, the same pixels carry a track id
    assert ((r0["track"] > 0) == (r1["track"] > 0)).all()
