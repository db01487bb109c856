"""CPU: pins `oracle/poly_oracle.py` (the restatement) against golden vectors produced by the
reference itself (`oracle/gen_golden.py`).  fp32 tolerance: 1e-5 relative-to-max for a single
stage (same arithmetic, different op order), bit-exact for the integer id maps."""
import json

import numpy as np
import pytest
import torch

import helpers as Hh
from oracle import poly_oracle as O

torch.set_grad_enabled(False)
TOL = 2e-5


def _meta(z):
    return json.loads(bytes(z["meta_json"]).decode())


def _weights(tag):
    if tag == "mini":
        z = Hh.load_golden("mini_weights.npz")
        return {k: torch.from_numpy(z[k]) for k in z.files}
    with open(Hh.GOLDEN + "/full_state_keys.json") as f:
        shapes = json.load(f)
    return Hh.seeded_fill(shapes, 1234)


def _iter_inputs(tag, m):
    cfg = m["cfg"]
    return Hh.iter_inputs(m["iseed"], m["B"], m["N"], cfg["C"], m["H"], m["W"])


@pytest.mark.parametrize("tag", ["mini", "full"])
def test_kernel_updator(tag):
    z = Hh.load_golden(f"{tag}_updator.npz")
    sd = _weights(tag)
    out = O.kernel_updator(sd, "roi_head.mask_head.0.kernel_update_conv",
                           torch.from_numpy(z["u"]), torch.from_numpy(z["k"]))
    assert Hh.rel_err(out, z["out"]) < TOL


@pytest.mark.parametrize("tag", ["mini", "full"])
def test_stage_teacher_forced(tag):
    z = Hh.load_golden(f"{tag}_iter.npz")
    m = _meta(z)
    sd = _weights(tag)
    inp = _iter_inputs(tag, m)
    if tag == "mini":   # stored inputs must equal the regenerated ones
        assert np.array_equal(z["in_x"], inp["x"].numpy())
    for s in range(m["cfg"]["S"]):
        r = O.update_stage(sd, f"roi_head.mask_head.{s}.", inp["x"],
                           torch.from_numpy(z[f"s{s}_in_k"]), torch.from_numpy(z[f"s{s}_in_m"]),
                           torch.from_numpy(z[f"s{s}_in_q"]), inp["dfe"], heads=m["cfg"]["heads"])
        for a, b in (("cls", "cls"), ("mask", "mask"), ("depth", "depth"), ("obj", "obj"), ("dobj", "dobj")):
            e = Hh.rel_err(r[a], z[f"s{s}_{b}"])
            assert e < TOL, (s, a, e)


@pytest.mark.parametrize("tag", ["mini", "full"])
def test_iter_free_running(tag):
    z = Hh.load_golden(f"{tag}_iter.npz")
    m = _meta(z)
    sd = _weights(tag)
    inp = _iter_inputs(tag, m)
    S = m["cfg"]["S"]
    out = O.iter_head_mask_preds(sd, S, inp["x"], inp["k0"], inp["m0"], inp["q0"], inp["dfe"],
                                 heads=m["cfg"]["heads"], prefix="roi_head.mask_head.")
    # free running: the hard threshold can amplify rounding; on these fixtures it does not
    assert Hh.rel_err(out["obj"], z["final_obj"]) < 1e-4
    assert Hh.rel_err(out["cls"], z["final_cls"]) < 1e-4
    assert Hh.rel_err(out["mask"], z[f"s{S - 1}_mask"]) < 1e-4
    assert Hh.rel_err(out["mask_up"], z["mask_up"]) < 1e-4
    assert Hh.rel_err(out["depth_up"], z["depth_up"]) < 1e-4


@pytest.mark.parametrize("tag", ["mini", "full"])
def test_kernel_head(tag):
    z = Hh.load_golden(f"{tag}_khead.npz")
    m = _meta(z)
    cfg = m["cfg"]
    sd = _weights(tag)
    feats = Hh.neck_inputs(m["nseed"], m["B"], cfg["C"], m["H"], m["W"])
    r = O.kernel_head_post_neck(sd, *feats, cfg["n_thing"], cfg["n_thing"] + cfg["n_stuff"],
                                cfg["groups"], prefix="rpn_head.")
    assert not r["depth_proposal"].is_contiguous()      # stride-0 expand view like the reference
    B, N = m["B"], m["N"]
    for k in ("x_feats", "mask_preds", "seg_preds", "depth_feats", "depth_pred"):
        assert Hh.rel_err(r[k], z[k]) < TOL, k
    assert Hh.rel_err(r["proposal_feats"].reshape(B, N, -1), z["proposal_feats"]) < TOL
    assert Hh.rel_err(r["depth_proposal"].reshape(B, N, -1), z["depth_proposal"]) < TOL


def _info_equal(a, b):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert x["id"] == y["id"] and x["isthing"] == y["isthing"] and x["category_id"] == y["category_id"]
        if x["isthing"]:
            assert x["instance_id"] == y["instance_id"] and abs(x["score"] - y["score"]) < 1e-6
        else:
            assert x["area"] == y["area"]


@pytest.mark.parametrize("case", ["a", "b", "c", "d", "e"])
def test_merge_crafted(case):
    """integer id map: bit-exact against the reference's get_panoptic on crafted inputs
    (duplicated query, equal scores, rescale + crop geometries; d, e: non-integer scale factors in both
    resampling steps with ori_shape != img_shape)."""
    z = Hh.load_golden("merge.npz" if case in "abc" else "merge2.npz")
    cfg = Hh.FULL
    h, w, bh, bw, oh, ow = [int(v) for v in z[f"{case}_meta"]]
    meta = Hh.img_meta(h, w, pad_to=(bh, bw), ori=(oh, ow))
    pan, info, d_basic, d_final = O.get_panoptic(
        torch.from_numpy(z[f"{case}_cls"]), torch.from_numpy(z[f"{case}_mask_up"]),
        torch.from_numpy(z[f"{case}_depth_up"]), torch.from_numpy(z[f"{case}_depth_init_up"]),
        meta, cfg["Nq"], cfg["n_thing"], cfg["Nq"])
    assert pan.dtype == np.int32 and np.array_equal(pan, z[f"{case}_pan"])
    _info_equal(info, json.loads(bytes(z[f"{case}_info"]).decode()))
    assert np.array_equal(d_basic, z[f"{case}_depth_basic"])
    assert np.array_equal(d_final, z[f"{case}_depth_final"])


@pytest.mark.parametrize("tag", ["mini", "full"])
def test_whole_path_panoptic(tag):
    z = Hh.load_golden(f"{tag}_panoptic.npz")
    m = _meta(z)
    cfg = m["cfg"]
    sd = _weights(tag)
    feats = Hh.neck_inputs(m["nseed"], m["B"], cfg["C"], m["H"], m["W"])
    out = O.run_head(sd, feats, cfg["S"], cfg["n_thing"], cfg["n_thing"] + cfg["n_stuff"],
                     cfg["heads"], cfg["groups"])
    d0_up = O.upsample2x(out["kernel_head"]["depth_pred"])
    for b in range(m["B"]):
        meta = Hh.img_meta(m["H"] * 8, m["W"] * 8)
        pan, info, d_basic, d_final = O.get_panoptic(out["cls"][b], out["mask_up"][b], out["depth_up"][b],
                                                     d0_up[b], meta, cfg["Nq"], cfg["n_thing"], cfg["Nq"])
        assert np.array_equal(pan, z[f"pan{b}"])
        _info_equal(info, json.loads(bytes(z[f"info{b}"]).decode()))
        assert Hh.rel_err(d_basic, z[f"depth_basic{b}"]) < 1e-5
        assert Hh.rel_err(d_final, z[f"depth_final{b}"]) < 1e-4
    h, w, bh, bw, oh, ow = [int(v) for v in z["geo2_meta"]]
    meta = Hh.img_meta(h, w, pad_to=(bh, bw), ori=(oh, ow))
    pan, info, _, _ = O.get_panoptic(out["cls"][0], out["mask_up"][0], out["depth_up"][0], d0_up[0],
                                     meta, cfg["Nq"], cfg["n_thing"], cfg["Nq"])
    assert np.array_equal(pan, z["pan_geo2"])


def cfg1_case():
    """(meta, weights, post-neck inputs, golden) of tests/golden/cfg1.npz: BASELINE configs[0] at its exact shape"""
    z = Hh.load_golden("cfg1.npz")
    m = _meta(z)
    shapes = json.loads(bytes(z["keys_json"]).decode())
    sd = Hh.seeded_fill(shapes, m["wseed"])
    feats = Hh.neck_inputs(m["nseed"], m["B"], m["cfg"]["C"], m["H"], m["W"])
    return m, sd, feats, z


def test_cfg1_exact_shape_oracle_vs_reference():
    """cfg1 = one 256x512 frame (32x64 map), N = 100 + 11, S = 1, fp32 CPU: the restatement against the reference's
    KernelHead -> simple_test_mask_preds at exactly that shape"""
    m, sd, feats, z = cfg1_case()
    cfg = m["cfg"]
    out = O.run_head(sd, feats, 1, cfg["n_thing"], cfg["n_thing"] + cfg["n_stuff"], cfg["heads"], cfg["groups"])
    kh = out["kernel_head"]
    B, N = m["B"], m["N"]
    assert Hh.rel_err(kh["mask_preds"], z["kh_mask_preds"]) < TOL
    assert Hh.rel_err(kh["proposal_feats"].reshape(B, N, -1), z["kh_proposal"]) < TOL
    assert Hh.rel_err(kh["depth_pred"], z["kh_depth_pred"]) < TOL
    for k in ("obj", "cls", "mask"):
        assert Hh.rel_err(out[k], z[k]) < 1e-4, k
    assert Hh.rel_err(out["mask_up"][..., 0::3, 0::3], z["mask_up_s"]) < 1e-4
    assert Hh.rel_err(out["depth_up"][..., 0::3, 0::3], z["depth_up_s"]) < 1e-4
    assert abs(float(out["mask_up"].double().abs().sum()) / z["mask_up_sum"][1] - 1) < 1e-5
