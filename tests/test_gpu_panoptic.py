"""GPU: panoptic merge (a7).  Integer work is checked BIT-EXACT:
  * `from_probs`: the merge kernels on materialised full-resolution probability / depth maps against the
    oracle's merge on the very same maps (ids, areas, accept decisions, pasted ids and depths).
  * fused path = the PRODUCT path of `simple_test` (logits -> ids with on-the-fly sigmoid + two-step bilinear):
    against the reference's own golden id maps, `np.array_equal` -- 0 differing pixels on every committed fixture,
    including two with non-integer scale factors in both resampling steps (merge2.npz); segment lists and stuff
    areas must agree exactly."""
import json

import numpy as np
import pytest
import torch

import helpers as Hh
from oracle import poly_oracle as O
from polyphonicformer_amd import panoptic as Pn
from polyphonicformer_amd.registry import ConfigDict

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
CFG = Hh.FULL


class _Head:
    """the attributes get_panoptic reads from KernelUpdateIterHead"""
    merge_joint, num_proposals, num_thing_classes = True, CFG["Nq"], CFG["n_thing"]
    test_cfg = ConfigDict(max_per_img=CFG["Nq"], merge_stuff_thing=dict(overlap_thr=0.6, instance_score_thr=0.3))
    mask_head = [type("S", (), {"depth_act_mode": "sigmoid"})()]


def _case(z, c):
    h, w, bh, bw, oh, ow = [int(v) for v in z[f"{c}_meta"]]
    return (torch.from_numpy(z[f"{c}_cls"]), torch.from_numpy(z[f"{c}_mask_up"]), torch.from_numpy(z[f"{c}_depth_up"]),
            torch.from_numpy(z[f"{c}_depth_init_up"]), Hh.img_meta(h, w, pad_to=(bh, bw), ori=(oh, ow)))


def _golden(case):
    return Hh.load_golden("merge.npz" if case in "abc" else "merge2.npz")


@pytest.mark.parametrize("case", ["a", "b", "c", "d", "e"])
def test_merge_from_probs_bit_exact(gpu, case):
    z = _golden(case)
    cls, m_up, d_up, d0_up, meta = _case(z, case)
    q, lab, sc = O.select_segments(cls, CFG["Nq"], CFG["n_thing"], CFG["Nq"])
    q2, lab2, sc2 = Pn.select_segments(cls, CFG["Nq"], CFG["n_thing"], CFG["Nq"])
    assert torch.equal(q, q2) and torch.equal(lab, lab2) and torch.equal(sc, sc2)
    P = O.rescale(m_up[q].sigmoid(), meta)
    D = O.rescale(O.depth_act(d_up, "sigmoid"), meta)[q]
    D0 = O.rescale(O.depth_act(d0_up, "sigmoid"), meta)[0]
    pan_ref, info_ref, dfin_ref = O.merge_from_probs(P, D, sc, lab, D0, CFG["n_thing"])
    Ho, Wo = P.shape[-2:]
    geom = (Pn.C.c_int32 * 8)(0, 0, 0, 0, 0, 0, Ho, Wo)
    pan, info, d_basic, d_final = Pn.merge_device(P.to(gpu).contiguous(), D.to(gpu).contiguous(), D0.to(gpu).contiguous(),
                                                  sc, lab, geom, (Ho, Wo), CFG["n_thing"], 0.3, 0.6, from_probs=True)
    assert pan.dtype == np.int32 and np.array_equal(pan, pan_ref)
    assert info == info_ref
    assert np.array_equal(d_final, dfin_ref.numpy()) and np.array_equal(d_basic, D0.numpy())
    # and the oracle on these maps reproduces the reference's golden id map
    assert np.array_equal(pan_ref, z[f"{case}_pan"])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", ["a", "b", "c", "d", "e"])
def test_get_panoptic_fused_vs_reference_golden(gpu, case, dtype):
    z = _golden(case)
    cls, m_up, d_up, d0_up, meta = _case(z, case)
    if dtype == torch.bfloat16:      # bf16 logits (benchmark output dtype): the golden must be recomputed on them
        m_up, d_up = m_up.to(dtype), d_up.to(dtype)
        pan_ref, info_ref, dbas_ref, dfin_ref = O.get_panoptic(cls, m_up.float(), d_up.float(), d0_up, meta, CFG["Nq"],
                                                               CFG["n_thing"], CFG["Nq"])
    else:
        pan_ref, info_ref = z[f"{case}_pan"], json.loads(bytes(z[f"{case}_info"]).decode())
        dbas_ref, dfin_ref = z[f"{case}_depth_basic"], z[f"{case}_depth_final"]
    out = Pn.get_panoptic(_Head, cls.to(gpu), m_up.to(gpu), d_up.to(gpu), d0_up.to(gpu), meta)
    assert out[0] is None and out[1] is None
    pan, info = out[2]
    bad = int((pan != pan_ref).sum())
    print(f"fused merge case {case} {dtype}: {bad} of {pan.size} pixels differ from the reference id map")
    assert pan.dtype == np.int32 and pan.shape == pan_ref.shape
    assert bad == 0                                  # integer mask-id assignment is bit-exact (north_star)
    assert [(s["id"], s["isthing"], s["category_id"], s.get("instance_id")) for s in info] == \
           [(s["id"], s["isthing"], s["category_id"], s.get("instance_id")) for s in info_ref]
    for a, b in zip(info, info_ref):
        if not a["isthing"]:
            assert a["area"] == b["area"]
    assert Hh.rel_err(out[3], dbas_ref) < 1e-5
    assert np.abs(out[4] - dfin_ref).max() < 1e-3 * np.abs(dfin_ref).max()


@pytest.mark.parametrize("shape", [(24, 40, 0, 0), (17, 23, 3, 2), (8, 16, 1, 3), (1, 1, 0, 0), (64, 128, 0, 0)])
def test_argmax_x4_form_equals_the_generic_kernel(gpu, shape, monkeypatch):
    """the shipped geometry (identity second resize, exact x4 first) takes a kernel of its own (4 x 2 pixel blocks, ballot
    histograms): ids and both histograms must be the generic kernel's bit for bit, cropped maps and borders included"""
    from polyphonicformer_amd import _lib
    sh, sw, cut_h, cut_w = shape
    K = 70
    g = torch.Generator().manual_seed(sh * 131 + sw)
    act = torch.rand(K, sh, sw, generator=g)
    act[:, : sh // 2] = (act[:, : sh // 2] * 8).round() / 8              # exact ties and exact 0.5s
    act[3, 0, 0] = float("nan")
    sc = torch.rand(K, generator=g)
    sc[5] = sc[4]
    Ho, Wo = 4 * sh - cut_h, 4 * sw - cut_w
    geom = (Pn.C.c_int32 * 8)(sh, sw, 4 * sh, 4 * sw, Ho, Wo, Ho, Wo)
    lib = _lib.load()
    a, s = act.to(gpu).contiguous(), sc.to(gpu)
    outs = []
    for generic in (False, True):
        if generic:
            monkeypatch.setenv("PH_PAN_GENERIC", "1")
        ids = torch.full((Ho, Wo), -7, dtype=torch.int32, device=gpu)
        cnt = torch.empty((2, K), dtype=torch.int32, device=gpu)
        _lib.check(lib.ph_panoptic_argmax(_lib.ptr(a), _lib.ptr(s), K, geom, 0, _lib.ptr(ids), _lib.ptr(cnt), _lib.stream_ptr()), "argmax")
        outs.append((ids.cpu(), cnt.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert int(outs[0][1][0].sum()) == Ho * Wo and int(outs[0][0].min()) >= 0


@pytest.mark.parametrize("case", ["a", "b", "c", "d", "e"])
def test_device_select_and_device_merge_equal_the_host_form(gpu, case):
    """ph_panoptic_select = panoptic.select_segments (torch.topk / sort on the host) on the golden class scores, for a batch of two
    frames; panoptic.DeviceMerge (select -> activate -> argmax with no host step, then accept + paste) = get_panoptic_device:
    ids, segment lists and depth maps identical"""
    from polyphonicformer_amd import _lib
    z = _golden(case)
    cls, m_up, d_up, d0_up, meta = _case(z, case)
    # (the fixtures hold exactly equal scores, whose order torch.topk / sort leave open: a small ramp breaks the ties for this test)
    cls = (cls.double() * 0.98 + 1e-6 * torch.arange(cls.numel(), dtype=torch.float64).reshape(cls.shape)).float()
    cls2 = torch.stack([cls, cls.flip(0).roll(3, 1) * 0.5 + 0.25 * cls])               # a second, different frame
    N, L = cls.shape
    K = CFG["Nq"] + min(N - CFG["Nq"], L - CFG["n_thing"])
    lib = _lib.load()
    out = torch.full((2, 3, K), -1, dtype=torch.int32, device=gpu)
    c = cls2.to(gpu).contiguous()
    i32 = lambda off: Pn.C.c_void_p(out.data_ptr() + 4 * off)
    _lib.check(lib.ph_panoptic_select(_lib.ptr(c), N * L, 2, N, L, CFG["Nq"], CFG["n_thing"], CFG["Nq"], i32(0), i32(K), i32(2 * K), 3 * K,
                                      _lib.stream_ptr()), "select")
    o = out.cpu()
    for b in range(2):
        q, lab, sc = Pn.select_segments(cls2[b], CFG["Nq"], CFG["n_thing"], CFG["Nq"])
        assert len(q) == K and len(set(sc.tolist())) == K                              # no exact ties in the fixture
        assert torch.equal(o[b, 0].long(), q) and torch.equal(o[b, 1].long(), lab) and torch.equal(o[b, 2].view(torch.float32), sc)
    # the whole merge, two frames per launch
    mm, dd, d0 = (torch.stack([t, t.flip(-1)]).to(gpu) for t in (m_up, d_up, d0_up))
    dm = Pn.DeviceMerge(_Head, c, mm, dd, d0, meta)
    dm.begin(c, mm, dd, d0)
    dm.download()
    torch.cuda.synchronize()
    for b in range(2):
        pan, info, d_basic, d_final = dm.finish(b)
        pan2, info2, d_basic2, d_final2 = Pn.get_panoptic_device(_Head, c[b], mm[b], dd[b], d0[b], meta)
        assert torch.equal(pan, pan2) and info == info2 and torch.equal(d_basic, d_basic2) and torch.equal(d_final, d_final2)


def test_simple_test_whole_path_golden(gpu):
    """KernelHead.simple_test_rpn -> KernelUpdateIterHead.simple_test exactly as Polyphonic.simple_test
    (polyphonic_former.py:145-161) against the reference's golden panoptic outputs."""
    from test_gpu_parity import _full_weights, _iter_head, _kernel_head
    z = Hh.load_golden("full_panoptic.npz")
    m = json.loads(bytes(z["meta_json"]).decode())
    B, H, W = m["B"], m["H"], m["W"]
    weights = _full_weights()
    kh, ih = _kernel_head(weights, "fp32"), _iter_head(weights, m["cfg"]["S"], precision="fp32")
    feats = [f.to(gpu) for f in Hh.neck_inputs(m["nseed"], B, 256, H, W)]
    metas = [Hh.img_meta(H * 8, W * 8) for _ in range(B)]
    (pf, xf, mp, cs, seg, df, dp, dpr, aspp) = kh.simple_test_rpn(feats, metas)
    res = ih.simple_test(xf, pf, mp, cs, metas, depth_preds=dpr, depth_feats=df, depth_proposal=dp, imgs_whwh=None,
                         aspp_semantic=aspp, rescale=True)
    assert len(res) == B
    for b in range(B):
        assert res[b][0] is None and res[b][1] is None
        pan, info = res[b][2]
        # free running through a1 + 3 stages + merge in fp32 precision: identical ids on the committed fixture
        bad = int((pan != z[f"pan{b}"]).sum())
        print(f"whole path image {b}: {bad} of {pan.size} id-map pixels differ from the reference")
        assert pan.dtype == np.int32 and bad == 0
        ref_info = json.loads(bytes(z[f"info{b}"]).decode())
        assert [(s["id"], s["category_id"]) for s in info] == [(s["id"], s["category_id"]) for s in ref_info]
        assert Hh.rel_err(res[b][3], z[f"depth_basic{b}"]) < 1e-3
        assert Hh.rel_err(res[b][4], z[f"depth_final{b}"]) < 1e-3
    # second geometry: padded batch shape + different ori_shape
    h, w, bh, bw, oh, ow = [int(v) for v in z["geo2_meta"]]
    meta2 = Hh.img_meta(h, w, pad_to=(bh, bw), ori=(oh, ow))
    (pf, xf, mp, cs, seg, df, dp, dpr, aspp) = kh.simple_test_rpn([f[:1] for f in feats], [meta2])
    res2 = ih.simple_test(xf, pf, mp, cs, [meta2], depth_preds=dpr, depth_feats=df, depth_proposal=dp)
    assert np.array_equal(res2[0][2][0], z["pan_geo2"])
