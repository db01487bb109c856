"""TEST INFRASTRUCTURE ONLY -- generates `tests/golden/*.npz` by running the REFERENCE itself
(`/root/reference`, imported through `oracle/ref_loader.py`) on CPU in the build container.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py

The fixtures hold inputs/expected outputs only (data, never reference source).  Two families:

* ``mini_*``  -- reduced widths (C=32 ...), everything stored incl. weights; used to pin the CPU
  restatement (`oracle/poly_oracle.py`) against the reference.
* ``full_*``  -- the shipped Cityscapes widths (C=256, F=2048, 8 heads, N=100+11, S=3) on a small
  8x16 stride-8 map; weights and inputs are NOT stored -- `tests/helpers.py` regenerates them from
  the recorded seeds -- only the reference's outputs are.  These pin both the restatement and, on
  the GPU box, the HIP path.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.dont_write_bytecode = True

import ref_loader as R  # noqa: E402
import helpers as Hh  # noqa: E402

torch.set_grad_enabled(False)
torch.set_num_threads(4)
OUT = Hh.GOLDEN
WSEED, ISEED, NSEED = 1234, 77, 99


def np_(t):
    return t.detach().cpu().numpy().astype(np.float32)


def build(ns, cfg):
    ih = R.build_iter_head(ns, S=cfg["S"], N_thing_q=cfg["Nq"], C=cfg["C"], F=cfg["F"],
                           heads=cfg["heads"], n_thing=cfg["n_thing"], n_stuff=cfg["n_stuff"])
    kh = R.build_kernel_head(ns, N_thing_q=cfg["Nq"], C=cfg["C"], n_thing=cfg["n_thing"],
                             n_stuff=cfg["n_stuff"], groups=cfg["groups"])
    shapes = {"roi_head." + k: v.shape for k, v in ih.state_dict().items()}
    shapes.update({"rpn_head." + k: v.shape for k, v in kh.state_dict().items()})
    sd = Hh.seeded_fill(shapes, WSEED)
    ih.load_state_dict({k[len("roi_head."):]: v for k, v in sd.items() if k.startswith("roi_head.")})
    kh.load_state_dict({k[len("rpn_head."):]: v for k, v in sd.items() if k.startswith("rpn_head.")})
    return ih, kh, sd, shapes


def run_family(ns, cfg, tag, B, H, W, store_all):
    ih, kh, sd, shapes = build(ns, cfg)
    N = cfg["Nq"] + cfg["n_stuff"]
    C = cfg["C"]
    metas = [Hh.img_meta(H * 8, W * 8) for _ in range(B)]
    fx = {}

    # ---- (1) KernelUpdator alone (funcs/kernel_updator.py:55-93)
    g = torch.Generator().manual_seed(5)
    u = torch.randn(B, N, C, generator=g) * 3.0
    k = torch.randn(B, N, 1, C, generator=g)
    upd = ih.mask_head[0].kernel_update_conv(u, k).reshape(B, N, C)
    fx["updator"] = dict(u=np_(u), k=np_(k.reshape(B, N, C)), out=np_(upd))

    # ---- (2) per-stage teacher-forced + (3) S-stage free-running (kernel_update.py:356-401)
    inp = Hh.iter_inputs(ISEED, B, N, C, H, W)
    obj, m, q = inp["k0"], inp["m0"], inp["q0"]
    dpre = inp["depth_pred"].expand(-1, N, -1, -1)
    stage = {}
    for s in range(cfg["S"]):
        stage[f"s{s}_in_k"] = np_(obj.reshape(B, N, C))
        stage[f"s{s}_in_q"] = np_(q.reshape(B, N, C))
        stage[f"s{s}_in_m"] = np_(m)
        r = ih._mask_forward(s, inp["x"], obj, m, metas, dpre, q, inp["dfe"])
        obj, m, q = r["object_feats"], r["mask_preds"], r["depth_proposal"]
        stage[f"s{s}_cls"] = np_(r["cls_score"])
        stage[f"s{s}_mask"] = np_(r["mask_preds"])
        stage[f"s{s}_depth"] = np_(r["depth_preds"])
        stage[f"s{s}_obj"] = np_(r["object_feats"].reshape(B, N, C))
        stage[f"s{s}_dobj"] = np_(r["depth_proposal"].reshape(B, N, C))
    stage["mask_up"] = np_(r["scaled_mask_preds"])
    stage["depth_up"] = np_(r["scaled_depth_preds"])
    o4 = ih.simple_test_mask_preds(inp["x"], inp["k0"], inp["m0"], None, metas,
                                   depth_preds=inp["depth_pred"], depth_feats=inp["dfe"],
                                   depth_proposal=inp["q0"])
    stage["final_obj"] = np_(o4[0].reshape(B, N, C))
    stage["final_cls"] = np_(o4[1])
    assert np.array_equal(np_(o4[2]), stage[f"s{cfg['S'] - 1}_mask"])
    assert np.array_equal(np_(o4[3]), stage["mask_up"])
    fx["iter"] = stage

    # ---- (4) KernelHead post-neck (kernel_head.py:245-347)
    feats = Hh.neck_inputs(NSEED, B, C, H, W)
    (pf, xf, mp, _, seg, df, dp, dpr, _) = kh.simple_test_rpn(feats, metas)
    assert not dp.is_contiguous()
    fx["khead"] = dict(proposal_feats=np_(pf.reshape(B, N, C)), x_feats=np_(xf), mask_preds=np_(mp),
                       seg_preds=np_(seg), depth_feats=np_(df),
                       depth_proposal=np_(dp.reshape(B, N, C)), depth_pred=np_(dpr))

    # ---- (5) whole path incl. panoptic merge (polyphonic_former.py:145-161, kernel_update.py:421-535)
    res = ih.simple_test(xf, pf, mp, None, metas, depth_preds=dpr, depth_feats=df, depth_proposal=dp)
    pan = {}
    for b in range(B):
        pan[f"pan{b}"] = res[b][2][0].astype(np.int32)
        pan[f"info{b}"] = np.frombuffer(json.dumps(res[b][2][1]).encode(), dtype=np.uint8)
        pan[f"depth_basic{b}"] = res[b][3].astype(np.float32)
        pan[f"depth_final{b}"] = res[b][4].astype(np.float32)
    # a second geometry: padded batch_input_shape and a different ori_shape (rescale path)
    meta2 = Hh.img_meta(H * 8 - 4, W * 8 - 8, pad_to=(H * 8, W * 8), ori=(H * 12 - 6, W * 12 - 12))
    res2 = ih.simple_test(xf[:1], pf[:1], mp[:1], None, [meta2], depth_preds=dpr[:1],
                          depth_feats=df[:1], depth_proposal=dp[:1])
    pan["pan_geo2"] = res2[0][2][0].astype(np.int32)
    pan["info_geo2"] = np.frombuffer(json.dumps(res2[0][2][1]).encode(), dtype=np.uint8)
    pan["depth_basic_geo2"] = res2[0][3].astype(np.float32)
    pan["depth_final_geo2"] = res2[0][4].astype(np.float32)
    pan["geo2_meta"] = np.array([H * 8 - 4, W * 8 - 8, H * 8, W * 8, H * 12 - 6, W * 12 - 12])
    fx["panoptic"] = pan
    nseg = [len(res[b][2][1]) for b in range(B)]
    print(f"[{tag}] segments per image: {nseg}; geo2: {len(res2[0][2][1])}")

    meta = dict(cfg=cfg, B=B, H=H, W=W, N=N, wseed=WSEED, iseed=ISEED, nseed=NSEED,
                torch=torch.__version__)
    for name, d in fx.items():
        d = dict(d)
        d["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
        if store_all:
            if name in ("updator", "iter"):
                d.update({"in_" + k: np_(v) for k, v in inp.items() if k != "q0"})
                d["in_q0"] = np_(inp["q0"].reshape(B, N, C))
            if name in ("khead", "panoptic"):
                for i, f in enumerate(feats):
                    d[f"in_f{i}"] = np_(f)
        np.savez_compressed(os.path.join(OUT, f"{tag}_{name}.npz"), **d)
    if store_all:
        np.savez_compressed(os.path.join(OUT, f"{tag}_weights.npz"), **{k: np_(v) for k, v in sd.items()})
    return shapes


def blob_logits(g, K, h, w, sharp=6.0, rmin=0.04, rvar=0.10):
    """K soft blobs (logit > 0 inside an ellipse) at random centres -- structured masks so that
    several segments survive the merge, unlike random-weight network outputs."""
    ys = torch.arange(h).view(1, h, 1).float()
    xs = torch.arange(w).view(1, 1, w).float()
    cy = torch.rand(K, 1, 1, generator=g) * h
    cx = torch.rand(K, 1, 1, generator=g) * w
    ry = (rmin + rvar * torch.rand(K, 1, 1, generator=g)) * h
    rx = (rmin + rvar * torch.rand(K, 1, 1, generator=g)) * w
    d = ((ys - cy) / ry) ** 2 + ((xs - cx) / rx) ** 2
    return sharp * (1.0 - d) + 0.3 * torch.randn(K, h, w, generator=g)


def merge_fixture(ns, tag="merge", cases=None, seed0=400):
    """Crafted inputs straight into the reference's get_panoptic (kernel_update.py:421-535)."""
    cfg = Hh.FULL
    ih = R.build_iter_head(ns, S=1, N_thing_q=cfg["Nq"], C=32, F=64, heads=4,
                           n_thing=cfg["n_thing"], n_stuff=cfg["n_stuff"])
    N, L = cfg["Nq"] + cfg["n_stuff"], cfg["n_thing"] + cfg["n_stuff"]
    out = {}
    cases = cases or [  # (h2, w2, img_meta)  h2,w2 = stride-4 (x2-upsampled) logits size
        ("a", 24, 48, Hh.img_meta(96, 192)),
        ("b", 24, 48, Hh.img_meta(90, 180, pad_to=(96, 192), ori=(135, 270))),
        ("c", 16, 40, Hh.img_meta(64, 160, ori=(48, 120))),
    ]
    for ci, (nm, h2, w2, meta) in enumerate(cases):
        g = torch.Generator().manual_seed(seed0 + ci)
        m_up = blob_logits(g, N, h2, w2)
        m_up[cfg["Nq"]:] = blob_logits(g, cfg["n_stuff"], h2, w2, sharp=3.0, rmin=0.15, rvar=0.3)   # stuff: broad
        cls = torch.rand(N, L, generator=g)
        cls[:cfg["Nq"]] *= (torch.rand(cfg["Nq"], 1, generator=g) < 0.4).float() * 0.9 + 0.1
        # ties / duplicates: query 1 duplicates query 0 (same mask, same best score)
        m_up[1] = m_up[0]
        cls[1] = cls[0]
        d_up = torch.randn(N, h2, w2, generator=g)
        d0_up = torch.randn(1, h2, w2, generator=g)
        r = ih.get_panoptic(cls, m_up, ih.test_cfg, meta, depth_preds=d_up, depth_init=d0_up,
                            aspp_semantic=None)
        out[f"{nm}_cls"], out[f"{nm}_mask_up"] = np_(cls), np_(m_up)
        out[f"{nm}_depth_up"], out[f"{nm}_depth_init_up"] = np_(d_up), np_(d0_up)
        out[f"{nm}_meta"] = np.array(list(meta["img_shape"][:2]) + list(meta["batch_input_shape"])
                                     + list(meta["ori_shape"][:2]))
        out[f"{nm}_pan"] = r[2][0].astype(np.int32)
        out[f"{nm}_info"] = np.frombuffer(json.dumps(r[2][1]).encode(), dtype=np.uint8)
        out[f"{nm}_depth_basic"], out[f"{nm}_depth_final"] = r[3].astype(np.float32), r[4].astype(np.float32)
        print(f"[merge {nm}] segments: {len(r[2][1])}, void px: {(r[2][0] == 0).sum()}")
    np.savez_compressed(os.path.join(OUT, f"{tag}.npz"), **out)


def tracker_fixture():
    """the reference's QuasiDenseEmbedTracker replayed over synthetic clips (configs/polyphonic_video/
    poly_r50_cityscapes_1x.py:51-64 settings) -> golden integer ids per frame"""
    Tr = R.load_reference_tracker()
    cfg = dict(init_score_thr=0.35, obj_score_thr=0.3, match_score_thr=0.5, memo_tracklet_frames=5,
               memo_backdrop_frames=1, memo_momentum=0.8, nms_conf_thr=0.5, nms_backdrop_iou_thr=0.3,
               nms_class_iou_thr=0.7, with_cats=True, match_metric="bisoftmax")
    out = {"cfg_json": np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8)}
    for seed in (1, 2, 3):
        tr = Tr(**cfg)
        cnt = 1
        for f, bb, lab, emb in Hh.tracker_records(seed):
            if bb.shape[0] == 0:
                continue
            obb, olab, ids = tr.match(bboxes=bb, labels=lab, track_feats=emb, frame_id=cnt)
            cnt += 1
            ids = ids + 1
            ids[ids == -1] = 0
            out[f"s{seed}_f{f}_ids"] = ids.numpy().astype(np.int64)
            out[f"s{seed}_f{f}_bboxes"] = obb.numpy().astype(np.float32)
            out[f"s{seed}_f{f}_labels"] = olab.numpy().astype(np.int64)
        print(f"[tracker seed {seed}] tracklets created: {tr.num_tracklets}")
    np.savez_compressed(os.path.join(OUT, "tracker.npz"), **out)


def video_fixture():
    """mask -> box helpers and the track embedding head of the reference on a crafted frame (video.npz)"""
    V = R.load_reference_track_head()
    pan, info, feats, roi_feats = Hh.video_case()
    masks = torch.stack([torch.from_numpy(pan == s["id"]).float() for s in info])
    stat = V.batch_mask2boxlist([masks])[0]                      # polyphonic_former_video.py:413
    rois = V.bboxlist2roi([stat]).clamp(min=0.0)                  # :414-415
    ext = torch.tensor(V.tensor_mask2box(masks))                  # :386-389
    head = V.Head(num_convs=4, num_fcs=1, embed_channels=256, norm_cfg=dict(type="GN", num_groups=32))
    sd = Hh.seeded_fill(Hh.TRACK_HEAD_SHAPES, 4321)
    head.load_state_dict({k[len("track_head."):]: v for k, v in sd.items()})
    head.eval()
    assert {"track_head." + k: tuple(v.shape) for k, v in head.state_dict().items()} == Hh.TRACK_HEAD_SHAPES
    emb = head(roi_feats)
    np.savez_compressed(os.path.join(OUT, "video.npz"), stat_boxes=np_(stat), rois=np_(rois), extent_boxes=np_(ext),
                        embeds=np_(emb))
    print("[video] segments:", len(info), "embed norm:", float(emb.norm(dim=1).mean()))


def cfg1_fixture(ns):
    """BASELINE.json configs[0] at its EXACT shape: one 256x512 frame -> 32x64 stride-8 map, 100 + 11 queries, ONE update stage,
    fp32 on the CPU: KernelHead post-neck -> simple_test_mask_preds of the reference.  Weights = seeded_fill over the key set of
    the S = 1 heads (regenerated by the tests from `cfg1_state_keys` = the sorted key list), inputs = neck_inputs(NSEED + 1)."""
    cfg = dict(Hh.FULL, S=1)
    ih, kh, sd, shapes = build(ns, cfg)
    B, H, W, C = 1, 32, 64, cfg["C"]
    N = cfg["Nq"] + cfg["n_stuff"]
    metas = [Hh.img_meta(H * 8, W * 8)]
    feats = Hh.neck_inputs(NSEED + 1, B, C, H, W)
    (pf, xf, mp, _, seg, df, dp, dpr, _) = kh.simple_test_rpn(feats, metas)
    r = ih._mask_forward(0, xf, pf, mp, metas, dpr.expand(-1, N, -1, -1), dp, df)
    o4 = ih.simple_test_mask_preds(xf, pf, mp, None, metas, depth_preds=dpr, depth_feats=df, depth_proposal=dp)
    assert np.array_equal(np_(o4[2]), np_(r["mask_preds"])) and np.array_equal(np_(o4[3]), np_(r["scaled_mask_preds"]))
    meta = dict(cfg=cfg, B=B, H=H, W=W, N=N, wseed=WSEED, nseed=NSEED + 1, torch=torch.__version__)
    np.savez_compressed(
        os.path.join(OUT, "cfg1.npz"), meta_json=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8),
        keys_json=np.frombuffer(json.dumps({k: list(v) for k, v in sorted(shapes.items())}).encode(), dtype=np.uint8),
        kh_mask_preds=np_(mp), kh_proposal=np_(pf.reshape(B, N, C)), kh_depth_pred=np_(dpr),
        obj=np_(o4[0].reshape(B, N, C)), cls=np_(o4[1]), mask=np_(o4[2]),
        # the x2-upsampled maps: every third row / column (both phases of the period-2 bilinear stencil) + two checksums
        mask_up_s=np_(o4[3][..., 0::3, 0::3]), depth_up_s=np_(r["scaled_depth_preds"][..., 0::3, 0::3]),
        mask_up_sum=np.array([float(o4[3].double().sum()), float(o4[3].double().abs().sum())]),
        depth_up_sum=np.array([float(r["scaled_depth_preds"].double().sum()), float(r["scaled_depth_preds"].double().abs().sum())]))
    print("[cfg1] 32x64, N = 111, S = 1: mask std", float(o4[2].std()))


def main():
    os.makedirs(OUT, exist_ok=True)
    ns = R.load_reference()
    video_fixture()
    tracker_fixture()
    merge_fixture(ns)
    # round 2: non-integer scale factors in BOTH resampling steps, ori_shape != img_shape (merge2.npz)
    merge_fixture(ns, tag="merge2", seed0=500, cases=[
        ("d", 24, 48, Hh.img_meta(93, 187, pad_to=(100, 200), ori=(140, 281))),
        ("e", 19, 37, Hh.img_meta(70, 141, pad_to=(75, 150), ori=(53, 107))),
    ])
    cfg1_fixture(ns)
    run_family(ns, Hh.MINI, "mini", B=2, H=6, W=10, store_all=True)
    shapes = run_family(ns, Hh.FULL, "full", B=2, H=8, W=16, store_all=False)
    with open(os.path.join(OUT, "full_state_keys.json"), "w") as f:
        json.dump({k: list(v) for k, v in sorted(shapes.items())}, f, indent=0)
    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print("golden bytes:", tot)


if __name__ == "__main__":
    main()
