"""CPU: host-side logic of the package (no GPU, no compute calls into the library):
the C-ABI library loads and exports every symbol include/polyhead.h declares, the registry /
state_dict contract, weight packing round trips, and the merge accept loop."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

import helpers as Hh
from oracle import poly_oracle as O


def test_library_exports_every_declared_symbol():
    from polyphonicformer_amd import _lib
    from polyphonicformer_amd.build import build_library
    build_library()
    hdr = open(os.path.join(Hh.REPO, "include", "polyhead.h")).read()
    declared = set(re.findall(r"\b(ph_[a-z0-9_]+)\s*\(", hdr)) - {"ph_hw_padded", "ph_n_padded"}
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    lib.ph_version.restype = ctypes.c_int
    assert lib.ph_version() == 100
    assert ctypes.sizeof(_lib.StageLayout) == 8 * (2 * 13 + 2 * 30) + 8 + 4 + 4


def test_no_cpu_fallback():
    from polyphonicformer_amd import _lib, engine as E
    with pytest.raises(_lib.PolyheadError):
        E.ingest(torch.zeros(1, 256, 4, 4), _lib.PH_PREC_BF16)
    with pytest.raises(_lib.PolyheadError):
        E.upsample2x(torch.zeros(1, 1, 4, 4))


def _heads():
    from polyphonicformer_amd.registry import HEADS, TRANSFORMER_LAYER
    import polyphonicformer_amd.kernel_head, polyphonicformer_amd.kernel_update  # noqa: F401,E401
    import polyphonicformer_amd.kernel_update_head, polyphonicformer_amd.kernel_updator  # noqa: F401,E401
    import bench
    S = 3
    ih = HEADS.build(dict(type="KernelUpdateIterHead", num_stages=S, assign_stages=S, stage_loss_weights=[1] * S,
                          num_proposals=100, num_thing_classes=8, num_stuff_classes=11, do_panoptic=True,
                          mask_head=bench.stage_cfg(19, 8, 11, 2048), test_cfg=dict(max_per_img=100),
                          some_future_kwarg=1))                      # **kwargs tolerated like the reference
    kh = HEADS.build(dict(type="KernelHead", num_proposals=100, num_classes=19, num_thing_classes=8,
                          num_stuff_classes=11, cat_stuff_mask=True, feat_downsample_stride=2, feat_refine=False,
                          use_binary=True, proposal_feats_with_obj=True, kernel_init_std=1, conv_normal_init=True,
                          loss_seg=dict(type="FocalLoss", use_sigmoid=True), test_cfg=None, train_cfg=None,
                          localization_fpn=None))
    assert "KernelUpdator" in TRANSFORMER_LAYER and "KernelUpdateHead" in HEADS
    return ih, kh


def test_state_dict_keys_match_the_reference():
    ih, kh = _heads()
    with open(Hh.GOLDEN + "/full_state_keys.json") as f:
        ref = json.load(f)
    got = {"roi_head." + k: list(v.shape) for k, v in ih.state_dict().items()}
    got.update({"rpn_head." + k: list(v.shape) for k, v in kh.state_dict().items()})
    assert got == ref
    # with the shipped localization_fpn config the neck's parameters appear under rpn_head.localization_fpn.*
    from polyphonicformer_amd.registry import HEADS
    import bench  # noqa: F401
    neck = dict(type="SemanticFPNWrapper", in_channels=256, feat_channels=256, out_channels=256, start_level=0, end_level=3,
                upsample_times=2, positional_encoding=dict(type="SinePositionalEncoding", num_feats=128, normalize=True),
                cat_coors=False, cat_coors_level=3, fuse_by_cat=False, return_list=False, num_aux_convs=2,
                norm_cfg=dict(type="GN", num_groups=32, requires_grad=True))
    kh2 = HEADS.build(dict(type="KernelHead", num_proposals=100, num_classes=19, num_thing_classes=8, num_stuff_classes=11,
                           cat_stuff_mask=True, feat_downsample_stride=2, feat_refine=False, use_binary=True,
                           proposal_feats_with_obj=True, kernel_init_std=1, conv_normal_init=True,
                           loss_seg=dict(type="FocalLoss", use_sigmoid=True), localization_fpn=neck))
    with open(Hh.GOLDEN + "/neck_state_keys.json") as f:
        nref = json.load(f)["full"]
    got2 = {k[len("localization_fpn."):]: list(v.shape) for k, v in kh2.state_dict().items() if k.startswith("localization_fpn.")}
    assert got2 == nref
    assert kh.num_proposals == 100 and ih.mask_head[0].mask_upsample_stride == 2
    assert ih.mask_head[0].num_classes == 19 and ih.mask_head[0].loss_cls.use_sigmoid


def test_unsupported_configs_fail_loudly():
    from polyphonicformer_amd.registry import HEADS
    import bench
    cfg = bench.stage_cfg(19, 8, 11, 2048)
    cfg["conv_kernel_size"] = 3
    with pytest.raises(NotImplementedError):
        HEADS.build(cfg)
    ih, kh = _heads()
    with pytest.raises(ValueError):             # heads built without train_cfg cannot assign
        ih.forward_train(None, None, None, None, [], [], [], depth_proposal=torch.zeros(1, 1))
    with pytest.raises(ValueError):
        kh.forward_train(None, [], [], [])
    with pytest.raises(NotImplementedError):
        ih.aug_test(None, None, None)


def test_forward_train_trains_through_the_neck_and_has_no_cpu_path():
    """round 5 (VERDICT r04 #1c): with this package's SemanticFPNWrapper as localization_fpn, forward_train no longer refuses -- the
    neck runs its differentiable form (tests/test_gpu_neck_train.py) and is trained like the reference's; the `frozen_neck_ok`
    escape hatch is gone.  Without a GPU the call fails loudly in libpolyhead's own check (no CPU path), not in a training guard."""
    from polyphonicformer_amd.registry import HEADS
    from polyphonicformer_amd import _lib
    import bench  # noqa: F401
    neck = dict(type="SemanticFPNWrapper", in_channels=256, feat_channels=256, out_channels=256, start_level=0, end_level=3,
                upsample_times=2, positional_encoding=dict(type="SinePositionalEncoding", num_feats=128, normalize=True),
                cat_coors=False, cat_coors_level=3, fuse_by_cat=False, return_list=False, num_aux_convs=2,
                norm_cfg=dict(type="GN", num_groups=32, requires_grad=True))
    tc = dict(assigner=dict(type="MaskHungarianAssignerWithDepth", cls_cost=dict(type="FocalLossCost", weight=2.0),
                            dice_cost=dict(type="DiceCost", weight=4.0, pred_act=True),
                            mask_cost=dict(type="MaskCost", weight=1.0, pred_act=True)),
              sampler=dict(type="MaskPseudoSampler"), pos_weight=1)
    kh = HEADS.build(dict(type="KernelHead", num_proposals=100, num_classes=19, num_thing_classes=8, num_stuff_classes=11,
                          cat_stuff_mask=True, feat_downsample_stride=2, feat_refine=False, use_binary=True,
                          proposal_feats_with_obj=True, kernel_init_std=1, conv_normal_init=True,
                          loss_seg=dict(type="FocalLoss", use_sigmoid=True), localization_fpn=neck, train_cfg=tc))
    assert kh.localization_fpn.differentiable and not hasattr(kh, "frozen_neck_ok")
    fpn = [torch.zeros(1, 256, 8 >> i, 16 >> i) for i in range(4)]
    with pytest.raises(_lib.PolyheadError, match="GPU"):
        kh.forward_train(fpn, [], [], [])


def test_feat_transform_cfg_is_not_mutated():
    from polyphonicformer_amd.registry import HEADS
    import bench
    cfg = bench.stage_cfg(19, 8, 11, 2048)
    cfg["feat_transform_cfg"]["kernel_size"] = 1
    HEADS.build(cfg)
    assert cfg["feat_transform_cfg"]["kernel_size"] == 1      # the reference pops it (kernel_update_head.py:125)


def test_pack_fragments_round_trip_and_folding():
    from polyphonicformer_amd import _lib
    from polyphonicformer_amd.pack import pack_b_fragments, pack_stage, unpack_b_fragments
    w = torch.randn(48, 64, dtype=torch.float64)
    assert torch.equal(unpack_b_fragments(pack_b_fragments(w), 48, 64), w)
    with open(Hh.GOLDEN + "/full_state_keys.json") as f:
        sd = Hh.seeded_fill(json.load(f), 1234)
    pre = "roi_head.mask_head.1."
    wb, wf, lay = pack_stage(sd, pre, 19, _lib.PH_PREC_SPLIT)
    assert wb.shape[0] == 2 and wb.dtype == torch.int16 and lay.ffn_dim == 2048 and lay.num_classes == 19
    # folded dynamic_layer: (pooled x) W'^T + cnt v == dynamic_layer(pooled(feat_transform(x)))
    off = lay.w[0][_lib.W_IDX["DYN"]]
    flat = sum(wb[p, off:off + 512 * 256].view(torch.bfloat16).double() for p in range(2))
    Wf = unpack_b_fragments(flat, 512, 256)
    Wx = sd[pre + "feat_transform.conv.weight"].reshape(256, 256).double()
    Wd = sd[pre + "kernel_update_conv.dynamic_layer.weight"].double()
    assert (Wf - Wd @ Wx).abs().max() < 2e-5 * (Wd @ Wx).abs().max()
    v = wf[lay.v[0][_lib.V_IDX["DYN_CNT"]]:][:512].double()
    assert torch.allclose(v, Wd @ sd[pre + "feat_transform.conv.bias"].double(), atol=1e-6)


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_accept_loop_matches_reference_segments(case):
    """host logic of the merge on histograms computed from the oracle's maps -> the reference's segments_info"""
    from polyphonicformer_amd import panoptic as Pn
    z = Hh.load_golden("merge.npz")
    cfg = Hh.FULL
    h, w, bh, bw, oh, ow = [int(v) for v in z[f"{case}_meta"]]
    meta = Hh.img_meta(h, w, pad_to=(bh, bw), ori=(oh, ow))
    cls = torch.from_numpy(z[f"{case}_cls"])
    q, lab, sc = Pn.select_segments(cls, cfg["Nq"], cfg["n_thing"], cfg["Nq"])
    P = O.rescale(torch.from_numpy(z[f"{case}_mask_up"])[q].sigmoid(), meta)
    ids = (sc.view(-1, 1, 1) * P).argmax(0)
    area = torch.bincount(ids.flatten(), minlength=len(q)).numpy()
    orig = (P >= 0.5).flatten(1).sum(1).numpy()
    newid, info = Pn.accept_loop(sc, lab, area, orig, cfg["n_thing"], 0.3, 0.6)
    ref = json.loads(bytes(z[f"{case}_info"]).decode())
    assert len(info) == len(ref)
    for a, b in zip(info, ref):
        assert a["id"] == b["id"] and a["category_id"] == b["category_id"] and a["isthing"] == b["isthing"]
    pan = torch.from_numpy(newid)[ids].numpy().astype(np.int32)
    assert np.array_equal(pan, z[f"{case}_pan"])


def test_accept_loop_compares_the_score_threshold_in_fp32():
    """kernel_update.py:503 tests `total_scores[k] < merge_cfg.instance_score_thr`: a 0-dim fp32 tensor against a Python scalar,
    which torch evaluates in fp32 -- a thing whose score is exactly fp32(0.7) = 0.69999998... is NOT below 0.7 and is kept (ADVICE r04;
    a double-precision comparison would drop it).  The oracle's loop and the product's vectorised form agree with torch."""
    from polyphonicformer_amd import panoptic as Pn
    from oracle import poly_oracle as O
    sc = torch.tensor([0.9, 0.7, 0.7, 0.5], dtype=torch.float32)               # 0.7 -> 0.699999988 in fp32
    assert not bool(sc[1] < 0.7)                                               # torch's own answer: the reference's arithmetic
    lab = torch.tensor([0, 1, 9, 2])                                            # index 2 is stuff (>= 8 thing classes): not thresholded
    newid, info = Pn.accept_loop(sc, lab, np.array([10, 10, 10, 10]), np.array([10, 10, 10, 10]), 8, 0.7, 0.5)
    assert newid.tolist() == [1, 2, 3, 0] and [s["category_id"] for s in info] == [0, 1, 9]
    newid, _ = Pn.accept_loop(sc, lab, np.array([10, 10, 10, 10]), np.array([10, 10, 10, 10]), 8, 0.7000001, 0.5)
    assert newid.tolist() == [1, 0, 2, 0]                                       # a threshold above fp32(0.7): below it, rejected
    # the oracle's sequential loop on disjoint one-pixel masks gives the same ids
    P = torch.zeros(4, 1, 4)
    for k in range(4):
        P[k, 0, k] = 1.0
    pan, oinfo, _ = O.merge_from_probs(P, torch.zeros(4, 1, 4), sc, lab, torch.zeros(1, 4), 8, 0.7, 0.5)
    assert pan.reshape(-1).tolist() == [1, 2, 3, 0] and [s["category_id"] for s in oinfo] == [0, 1, 9]


def test_fp16_config_key_switches_both_heads():
    """the reference's only mixed-precision hook is the config's `fp16` key (tools/test.py:202-204: wrap_fp16_model): heads
    built from a detector config that carries it run at their fp16 grades, without it at the parity grade"""
    import torch
    from helpers import stage_cfg
    from polyphonicformer_amd.registry import build_heads_from_config
    import polyphonicformer_amd.kernel_head, polyphonicformer_amd.kernel_update  # noqa: F401,E401
    model = dict(
        type="PolyphonicFormer",
        rpn_head=dict(type="KernelHead", num_proposals=100, num_classes=19, num_thing_classes=8, num_stuff_classes=11,
                      cat_stuff_mask=True, feat_downsample_stride=2, feat_refine=False, use_binary=True, proposal_feats_with_obj=True,
                      localization_fpn=None, loss_seg=dict(type="FocalLoss", use_sigmoid=True)),
        roi_head=dict(type="KernelUpdateIterHead", num_stages=3, assign_stages=3, stage_loss_weights=[1, 1, 1], num_proposals=100,
                      num_thing_classes=8, num_stuff_classes=11, do_panoptic=True, mask_head=stage_cfg(256, 2048, 8, 19, 8, 11)),
        train_cfg=None, test_cfg=dict(rpn=None, rcnn=dict(max_per_img=100, mask_thr=0.5)))
    rpn, roi = build_heads_from_config(dict(model=model))
    assert rpn.precision == "fp32" and roi.precision == "fp32" and roi.output_dtype == torch.float32
    assert roi.test_cfg.max_per_img == 100
    rpn, roi = build_heads_from_config(dict(model=model, fp16=dict(loss_scale=512.)))
    assert rpn.precision == "fp16" and roi.precision == "fp16" and roi.output_dtype == torch.float16
    assert all(h.precision == "fp16" for h in roi.mask_head)


def test_param_versions_sees_in_place_updates_and_replaced_parameters():
    """_lib.param_versions (the cache key of the packed device weights) walks a cached module list: an in-place update bumps it,
    a replaced Parameter object is seen through the live `_parameters` dicts, and a deep copy gets its own list"""
    import copy
    from polyphonicformer_amd import _lib
    m = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.LayerNorm(4))
    v0 = _lib.param_versions(m)
    assert len(v0) == 4 and _lib.param_versions(m) == v0
    with torch.no_grad():
        m[0].weight.add_(1.0)
    v1 = _lib.param_versions(m)
    assert v1 != v0
    m[1].bias = torch.nn.Parameter(torch.zeros(4))
    with torch.no_grad():
        m[1].bias.add_(1.0)
        m[1].bias.add_(1.0)
    assert _lib.param_versions(m) != v1
    m2 = copy.deepcopy(m)
    with torch.no_grad():
        m2[0].weight.mul_(2.0)
    assert _lib.param_versions(m2) != _lib.param_versions(m)
    assert _lib._tree(m2)[0][1] is m2 and _lib._tree(m2)[1][1] is m2[0]
    # a REPLACED or ADDED sub-module is seen (the cached tree is keyed on the identity of every module's children),
    # and a Parameter shared by two modules is listed once, like nn.Module.named_parameters
    v2 = _lib.param_versions(m)
    old = set(map(id, _lib.named_params(m).values()))
    m[0] = torch.nn.Linear(4, 4)
    assert _lib.param_versions(m) != v2
    assert id(m[0].weight) in set(map(id, _lib.named_params(m).values())) - old
    m.add_module("extra", torch.nn.Linear(4, 2))
    assert "extra.weight" in _lib.named_params(m) and len(_lib.param_versions(m)) == 6
    m.extra.weight = m[0].weight if m.extra.weight.shape == m[0].weight.shape else m.extra.weight
    tied = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.Linear(4, 4))
    tied[1].weight = tied[0].weight
    assert list(_lib.named_params(tied)) == [n for n, _ in tied.named_parameters()]
