"""GPU: the BASELINE.json configurations at their FULL sizes against the CPU oracle.

  * cfg2  1024x2048 -> 128x256, N = 153 (100 + 53), L = 133, S = 3: every stage teacher-forced from the oracle's own
    stage inputs (one frame, ~1 s of CPU), both precisions, plus the IDENTICAL-INPUTS form of the test: the oracle is
    fed the bf16-rounded feature maps the device consumes (that is what cfg2's "bf16" inputs are), so the measured
    error is the path's arithmetic and not the input rounding.
  * cfg3  same map size, N = 111 (the shipped video head), two frames.
  * cfg5  1242x375 padded to 1248x384 -> 48x156, N = 253 (200 + 53), S = 3: teacher-forced per stage AND
    `simple_test_mask_preds` free running (flip rate of the hard threshold inside the recurrence reported and bounded).

Tolerances: see TOL_IDENT / TOL below and tests/test_gpu_parity.py's header."""
import pytest
import torch

import bench
import helpers as Hh
from oracle import poly_oracle as O

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

TOL = {"fp32": 1e-3, "bf16": 3e-2, "mixed": 3e-3, "mixed16": 3e-3, "fp16": 1e-3}
# identical (16-bit-rounded) feature inputs on both sides: north_star's 1e-3 for the modes that claim it
TOL_IDENT = {"fp32": 1e-3, "mixed": 1e-3, "mixed16": 1e-3, "fp16": 1e-3, "bf16": 3e-2}
# second, element-wise criterion of every teacher-forced output: |a - b| <= tol * |b| + ATOL_FRAC * tol * max|b| on EVERY element.
# An entry far below the tensor's maximum must then be within ATOL_FRAC * tol of the maximum (tighter than `rel_err < tol` allows),
# a large entry may use its own magnitude -- so this assertion can fail where the max-normalised one passes and vice versa.
ATOL_FRAC = 0.75
# free running (three stages of accumulated arithmetic, the oracle following the device's hard masks): the same mixed criterion with
# 0.95 -- an error at a SMALL entry has to stay below 0.95 tol of the maximum, a large entry may use its own magnitude
ATOL_FRAC_FREE = 0.95
PLANE_DT = {"bf16": torch.bfloat16, "mixed": torch.bfloat16, "mixed16": torch.bfloat16, "fp16": torch.float16, "fp32": None}

CFG2 = dict(H=128, W=256, Nq=100, n_thing=80, n_stuff=53, S=3, F=2048)
CFG3 = dict(H=128, W=256, Nq=100, n_thing=8, n_stuff=11, S=3, F=2048)
CFG5 = dict(H=48, W=156, Nq=200, n_thing=80, n_stuff=53, S=3, F=2048)


def _head_and_sd(wl, precision, gpu, out_dtype=torch.float32, seed=0):
    head = bench.build_head(wl, precision, out_dtype, gpu, seed=seed)
    sd = {k: v.detach().cpu() for k, v in head.state_dict().items()}
    return head, sd


def _teacher_forced(head, sd, wl, inp, gpu, tol, feats_dtype=None):
    """every stage on the oracle's own stage inputs; returns {(stage, output): rel err}"""
    B = inp["x"].shape[0]
    N = wl["Nq"] + wl["n_stuff"]
    ref = O.iter_head_mask_preds(sd, wl["S"], inp["x"], inp["k0"], inp["m0"], inp["q0"], inp["dfe"], return_stages=True)
    x, dfe = inp["x"].to(gpu), inp["dfe"].to(gpu)
    if feats_dtype is not None:
        x, dfe = x.to(feats_dtype), dfe.to(feats_dtype)
    k, q, m = inp["k0"].reshape(B, N, 256), inp["q0"].reshape(B, N, 256), inp["m0"]
    errs, worst_at = {}, 0.0
    for s in range(wl["S"]):
        r = ref["stages"][s]
        cls, nm, obj, nd, dobj = head.mask_head[s](x, k.to(gpu).reshape(B, N, 256, 1, 1), m.to(gpu),
                                                    depth_proposal=q.to(gpu).reshape(B, N, 256, 1, 1), depth_feats=dfe)
        got = dict(cls=cls, mask=nm, obj=obj.reshape(B, N, 256), depth=nd, dobj=dobj.reshape(B, N, 256))
        for name, t in got.items():
            errs[(s, name)] = Hh.rel_err(t.float().cpu(), r[name])
            at = Hh.needed_atol(t.float().cpu(), r[name], tol)
            worst_at = max(worst_at, at)
            assert at < ATOL_FRAC * tol, (s, name, at)
        k, q, m = r["obj"], r["dobj"], r["mask"]                 # teacher forcing: the oracle's outputs feed the next stage
    worst = max(errs.values())
    print("teacher-forced rel err:", {f"s{s}.{n}": f"{v:.1e}" for (s, n), v in errs.items()},
          f"| needed atol / max|b| at rtol = {tol:g}: {worst_at:.1e}")
    assert worst < tol, (worst, errs)
    return errs


@pytest.mark.parametrize("precision", ["fp32", "bf16", "mixed", "mixed16", "fp16"])
def test_cfg2_full_size_stage_vs_oracle(gpu, precision):
    """cfg2 at its full size against the oracle (fp32 NCHW inputs, as the reference API hands them).
    'mixed' rounds the fp32 inputs to one bf16 plane (its 1e-3 claim is for bf16 inputs, next test): 3e-3 here;
    'fp16' rounds them to fp16 (2^-12) and stays inside 1e-3 even against the unrounded inputs."""
    head, sd = _head_and_sd(CFG2, precision, gpu)
    inp = bench.synth_inputs(CFG2, 1, seed=11)
    _teacher_forced(head, sd, CFG2, inp, gpu, TOL[precision])


@pytest.mark.parametrize("precision", ["fp32", "bf16", "mixed", "mixed16", "fp16"])
def test_cfg2_identical_16bit_inputs(gpu, precision):
    """both sides consume the SAME 16-bit-rounded feature maps (bf16 = cfg2's input dtype; fp16 for the 'fp16' mode): what
    is left is the arithmetic error of the path itself.  The modes that claim north_star's 1e-3 are gated at 1e-3 here."""
    head, sd = _head_and_sd(CFG2, precision, gpu)
    inp = bench.synth_inputs(CFG2, 1, seed=12)
    rd = torch.float16 if precision == "fp16" else torch.bfloat16
    inp["x"], inp["dfe"] = inp["x"].to(rd).float(), inp["dfe"].to(rd).float()
    _teacher_forced(head, sd, CFG2, inp, gpu, TOL_IDENT[precision], feats_dtype=PLANE_DT[precision])   # 16-bit NCHW = planes


@pytest.mark.parametrize("precision,out_dtype", [("mixed", torch.float16), ("mixed16", torch.float16), ("fp16", torch.float16), ("mixed", torch.float32)])
def test_cfg2_headline_function_identical_inputs(gpu, monkeypatch, precision, out_dtype):
    """`simple_test_mask_preds` itself (what bench.py times), S = 3 free running, in the modes that claim 1e-3: 16-bit
    feature tensors in (as the bench hands them), 16-bit logits out; the oracle gets the same rounded features.  Free
    running through three hard thresholds, a logit within rounding of the threshold flips a pixel and moves a whole
    feature vector (the parity contract is per stage, teacher forced -- the tests above); here the flip rate
    is the bounded quantity (measured: fp32 mode 1.5e-4 at cfg5, 'mixed' 1.2e-4, 'fp16' 9e-4 at cfg2), 1e-3 is
    asserted whenever no pixel flipped."""
    wl = CFG2
    # the bench's launches: at 24 frames per part the final stage is the fused conv + x2 upsample kernel (chosen from B * H >= 512);
    # forced here so that ONE frame takes the same kernels
    monkeypatch.setenv("PH_CONV_UP2", "1")
    head, sd = _head_and_sd(wl, precision, gpu, out_dtype=out_dtype)
    inp = bench.synth_inputs(wl, 1, seed=15)
    rd = PLANE_DT[precision]
    inp["x"], inp["dfe"] = inp["x"].to(rd).float(), inp["dfe"].to(rd).float()
    if out_dtype != torch.float32:       # round 5: the bench hands the initial mask logits over in the mode's 16-bit logit format too
        inp["m0"] = inp["m0"].to(out_dtype).float()
    N = wl["Nq"] + wl["n_stuff"]
    ref = O.iter_head_mask_preds(sd, wl["S"], inp["x"], inp["k0"], inp["m0"], inp["q0"], inp["dfe"])
    g = {k: v.to(gpu) for k, v in inp.items()}
    if out_dtype != torch.float32:
        g["m0"] = g["m0"].to(out_dtype)
    obj, cls, mask, mask_up = head.simple_test_mask_preds(g["x"].to(rd), g["k0"].reshape(1, N, 256, 1, 1), g["m0"], None,
                                                          [Hh.img_meta(1024, 2048)], depth_feats=g["dfe"].to(rd),
                                                          depth_proposal=g["q0"].reshape(1, N, 256, 1, 1))
    assert mask.dtype == out_dtype and mask_up.dtype == out_dtype and mask_up.shape == (1, N, 256, 512)
    plan = next(iter(head._plans.values()))
    assert plan.fused_up == (precision in ("mixed16", "fp16") and out_dtype == torch.float16)      # `mixed` has two kernel planes
    flips = ((mask.float().cpu() > 0) != (ref["mask"] > 0)).float().mean().item()
    e = {n: Hh.rel_err(t.float().cpu(), r) for n, t, r in (("obj", obj.reshape(1, N, 256), ref["obj"]), ("cls", cls, ref["cls"]),
                                                            ("mask", mask, ref["mask"]), ("mask_up", mask_up, ref["mask_up"]),
                                                            ("depth_up", plan.depth_up, ref["depth_up"]))}
    print(f"cfg2 simple_test_mask_preds {precision}/{out_dtype}: flip rate {flips:.2e}, rel err vs the free-running oracle", {k: f"{v:.1e}" for k, v in e.items()})
    assert flips < (1e-3 if precision == "mixed" else 5e-3)
    # no blanket 5e-2 once a pixel flipped.  The same call again, recording the hard masks every stage of the
    # DEVICE run pooled with (DecodePlan.debug_bits); the oracle then follows those decisions (hard_masks=) instead of its own
    # thresholds, so what is compared is the arithmetic of the timed function through all three stages -- at 1e-3 on every
    # output, whatever flipped -- and the per-stage flip rates are the separately bounded quantity.
    plan.debug_bits = []
    obj, cls, mask, mask_up = head.simple_test_mask_preds(g["x"].to(rd), g["k0"].reshape(1, N, 256, 1, 1), g["m0"], None,
                                                          [Hh.img_meta(1024, 2048)], depth_feats=g["dfe"].to(rd),
                                                          depth_proposal=g["q0"].reshape(1, N, 256, 1, 1))
    torch.cuda.synchronize()
    dev_bits, plan.debug_bits = plan.debug_bits, None
    assert len(dev_bits) == wl["S"]
    HW = wl["H"] * wl["W"]
    import numpy as np
    hard = []
    for b in dev_bits:
        u = np.unpackbits(b.cpu().numpy().view("uint32").view("uint8"), axis=-1, bitorder="little")[:, :N, :HW]
        hard.append(torch.from_numpy(u.astype("float32")).reshape(1, N, wl["H"], wl["W"]))
    refs = O.iter_head_mask_preds(sd, wl["S"], inp["x"], inp["k0"], inp["m0"], inp["q0"], inp["dfe"], return_stages=True)
    stage_flips = [float((hard[0] != O.binarize(inp["m0"])).float().mean())] + \
                  [float((hard[s + 1] != O.binarize(refs["stages"][s]["mask"])).float().mean()) for s in range(wl["S"] - 1)]
    assert stage_flips[0] == 0.0                                   # binarising the given logits is exact
    refc = O.iter_head_mask_preds(sd, wl["S"], inp["x"], inp["k0"], inp["m0"], inp["q0"], inp["dfe"], hard_masks=hard)
    plan = next(iter(head._plans.values()))
    ec = {n: Hh.rel_err(t.float().cpu(), r) for n, t, r in (("obj", obj.reshape(1, N, 256), refc["obj"]), ("cls", cls, refc["cls"]),
                                                             ("mask", mask, refc["mask"]), ("mask_up", mask_up, refc["mask_up"]),
                                                             ("depth_up", plan.depth_up, refc["depth_up"]))}
    print(f"   hard masks of the device run: flip rate per stage input {[f'{f:.1e}' for f in stage_flips]}; "
          f"rel err vs the oracle on the same hard masks", {k: f"{v:.1e}" for k, v in ec.items()})
    assert max(stage_flips) < (1e-3 if precision == "mixed" else 5e-3)
    assert max(ec.values()) < 1e-3, ec
    at = {n: Hh.needed_atol(t.float().cpu(), r, 1e-3) for n, t, r in (("obj", obj.reshape(1, N, 256), refc["obj"]), ("cls", cls, refc["cls"]),
                                                                      ("mask", mask, refc["mask"]), ("mask_up", mask_up, refc["mask_up"]),
                                                                      ("depth_up", plan.depth_up, refc["depth_up"]))}
    print("   element-wise: needed atol / max|b| at rtol = 1e-3", {k: f"{v:.1e}" for k, v in at.items()})
    assert max(at.values()) < ATOL_FRAC_FREE * 1e-3, at


@pytest.mark.parametrize("precision", ["fp32", "bf16", "fp16", "mixed16"])
def test_cfg3_video_head_size_two_frames(gpu, precision):
    """cfg3's head (N = 100 + 11) at full map size, two frames, teacher forced -- also in the grades the video legs run
    (`fp16`; `mixed16` against fp32 inputs is gated at its input-rounding tolerance, as at cfg2)"""
    head, sd = _head_and_sd(CFG3, precision, gpu, seed=2)
    inp = bench.synth_inputs(CFG3, 2, seed=13)
    _teacher_forced(head, sd, CFG3, inp, gpu, TOL[precision])


@pytest.mark.parametrize("precision", ["fp32", "bf16", "fp16", "mixed16", "mixed"])
def test_cfg5_teacher_forced_and_free_running(gpu, precision):
    """cfg5's shape (48x156: HW = 7488 is not a multiple of 128, N = 253 -> 8 row tiles), all three stages"""
    wl = CFG5
    head, sd = _head_and_sd(wl, precision, gpu, seed=5, out_dtype=torch.float16 if precision == "fp16" else torch.float32)
    inp = bench.synth_inputs(wl, 1, seed=14)
    _teacher_forced(head, sd, wl, inp, gpu, TOL[precision])
    # free running through simple_test_mask_preds, on IDENTICAL inputs: the 16-bit grades get features already rounded to
    # their plane format (what cfg5 "fp16" means; for `mixed` / `mixed16` / `bf16` the bf16 plane the ingest would produce),
    # the oracle gets the same rounded values.  The run records the hard masks every stage pooled with, the oracle follows
    # them: every output at the mode's per-stage tolerance (1e-3 for all but the all-bf16 grade) whatever pixel flipped
    # (no blanket 5e-2, a real bound for mixed16), the per-stage flip rates bounded separately.
    N = wl["Nq"] + wl["n_stuff"]
    rd = PLANE_DT[precision]
    if rd is not None:
        inp["x"], inp["dfe"] = inp["x"].to(rd).float(), inp["dfe"].to(rd).float()
    g = {k: v.to(gpu) for k, v in inp.items()}
    metas = [Hh.img_meta(375, 1242, pad_to=(384, 1248))]
    plan = head._plan(1, N, wl["H"], wl["W"], gpu)
    plan.debug_bits = []
    obj, cls, mask, mask_up = head.simple_test_mask_preds(g["x"], g["k0"].reshape(1, N, 256, 1, 1), g["m0"], None, metas,
                                                          depth_feats=g["dfe"], depth_proposal=g["q0"].reshape(1, N, 256, 1, 1))
    torch.cuda.synchronize()
    assert mask_up.shape == (1, N, 96, 312) and obj.shape == (1, N, 256, 1, 1)
    plan = next(iter(head._plans.values()))
    hard, plan.debug_bits = Hh.unpack_hard_masks(plan.debug_bits, N, wl["H"], wl["W"]), None
    assert len(hard) == wl["S"]
    refs = O.iter_head_mask_preds(sd, wl["S"], inp["x"], inp["k0"], inp["m0"], inp["q0"], inp["dfe"], return_stages=True)
    stage_flips = [float((hard[0] != O.binarize(inp["m0"])).float().mean())] + \
                  [float((hard[s + 1] != O.binarize(refs["stages"][s]["mask"])).float().mean()) for s in range(wl["S"] - 1)]
    refc = O.iter_head_mask_preds(sd, wl["S"], inp["x"], inp["k0"], inp["m0"], inp["q0"], inp["dfe"], hard_masks=hard)
    e = {n: Hh.rel_err(t.float().cpu(), r) for n, t, r in (("obj", obj.reshape(1, N, 256), refc["obj"]), ("cls", cls, refc["cls"]),
                                                            ("mask", mask, refc["mask"]), ("mask_up", mask_up, refc["mask_up"]),
                                                            ("depth_up", plan.depth_up, refc["depth_up"]))}
    print(f"cfg5 free-running {precision}: flip rate per stage input {[f'{f:.1e}' for f in stage_flips]}, "
          f"rel err vs the oracle on the device's hard masks", {k: f"{v:.1e}" for k, v in e.items()})
    assert stage_flips[0] == 0.0                                   # binarising the given logits is exact
    assert max(stage_flips) < {"fp32": 1e-3, "bf16": 5e-2}.get(precision, 5e-3)
    assert max(e.values()) < TOL_IDENT[precision], e
    tol = TOL_IDENT[precision]
    at = {n: Hh.needed_atol(t.float().cpu(), r, tol) for n, t, r in (("obj", obj.reshape(1, N, 256), refc["obj"]), ("cls", cls, refc["cls"]),
                                                                     ("mask", mask, refc["mask"]), ("mask_up", mask_up, refc["mask_up"]),
                                                                     ("depth_up", plan.depth_up, refc["depth_up"]))}
    print(f"   element-wise: needed atol / max|b| at rtol = {tol:g}", {k: f"{v:.1e}" for k, v in at.items()})
    assert max(at.values()) < ATOL_FRAC_FREE * tol, at


def test_api_outputs_survive_the_next_call(gpu):
    """the reference returns fresh tensors; results kept from frame t must not change when frame
    t + 1 is decoded with the same shapes (e.g. the key / reference frames of polyphonic_former_video.py:208-242)"""
    wl = dict(H=16, W=32, Nq=100, n_thing=8, n_stuff=11, S=2, F=2048)
    head, _ = _head_and_sd(wl, "bf16", gpu)
    N = wl["Nq"] + wl["n_stuff"]
    metas = [Hh.img_meta(128, 256)]
    outs = []
    for seed in (21, 22):
        g = {k: v.to(gpu) for k, v in bench.synth_inputs(wl, 1, seed=seed).items()}
        outs.append(head.simple_test_mask_preds(g["x"], g["k0"].reshape(1, N, 256, 1, 1), g["m0"], None, metas,
                                                depth_feats=g["dfe"], depth_proposal=g["q0"].reshape(1, N, 256, 1, 1)))
        if seed == 21:
            keep = [t.clone() for t in outs[0]]
    torch.cuda.synchronize()
    for a, b in zip(outs[0], keep):
        assert torch.equal(a, b)                                  # frame t's tensors are untouched
    assert not torch.equal(outs[0][2], outs[1][2])                # and frame t+1 really produced something else
    # KernelHead: the 9-tuple of call 1 survives call 2, and its hand-off still decodes to the same result afterwards
    import polyphonicformer_amd.kernel_head  # noqa: F401
    from test_gpu_parity import _full_weights, _iter_head, _kernel_head
    w = _full_weights()
    kh, ih = _kernel_head(w, "fp32"), _iter_head(w, 2, precision="fp32")
    m8 = [Hh.img_meta(64, 128)]
    r1 = kh.simple_test_rpn([f.to(gpu) for f in Hh.neck_inputs(31, 1, 256, 8, 16)], m8)
    snap = [t.clone() for t in (r1[0], r1[1], r1[2], r1[4], r1[5], r1[7])]
    d1 = ih.simple_test_mask_preds(r1[1], r1[0], r1[2], r1[3], m8, depth_preds=r1[7], depth_feats=r1[5], depth_proposal=r1[6])
    d1 = [t.clone() for t in d1]
    r2 = kh.simple_test_rpn([f.to(gpu) for f in Hh.neck_inputs(32, 1, 256, 8, 16)], m8)
    ih.simple_test_mask_preds(r2[1], r2[0], r2[2], r2[3], m8, depth_preds=r2[7], depth_feats=r2[5], depth_proposal=r2[6])
    for a, b in zip((r1[0], r1[1], r1[2], r1[4], r1[5], r1[7]), snap):
        assert torch.equal(a, b)
    d1b = ih.simple_test_mask_preds(r1[1], r1[0], r1[2], r1[3], m8, depth_preds=r1[7], depth_feats=r1[5], depth_proposal=r1[6])
    for a, b in zip(d1, d1b):
        assert torch.equal(a, b)


@pytest.mark.parametrize("precision", ["fp32", "fp16", "bf16"])
def test_cfg2_full_size_kernel_head_vs_oracle(gpu, precision):
    """a1 at cfg2's full stride-8 size (128 x 256, one frame, 100 + 11 rows) against the oracle, in its three grades: GroupNorm
    statistics over 32768 pixels, the static convs, x = sem + loc.  'fp16' is the grade that feeds the decode's fp16 mode."""
    from test_gpu_parity import _full_weights, _kernel_head
    w = _full_weights()
    sd = {k[len("rpn_head."):]: v for k, v in w.items() if k.startswith("rpn_head.")}
    H, W = 128, 256
    feats = Hh.neck_inputs(57, 1, 256, H, W)
    ref = O.kernel_head_post_neck(sd, *feats, 8, 19, 32)
    kh = _kernel_head(w, precision)
    out = kh.simple_test_rpn([f.to(gpu) for f in feats], [Hh.img_meta(H * 8, W * 8)])
    tol = {"fp32": 1e-3, "fp16": 1e-3, "bf16": 3e-2}[precision]
    errs = {name: Hh.rel_err(t.float().cpu(), ref[name]) for name, t in (("x_feats", out[1]), ("mask_preds", out[2]), ("seg_preds", out[4]),
                                                                         ("depth_feats", out[5]), ("depth_pred", out[7]))}
    print("KernelHead full size", precision, {k: f"{v:.1e}" for k, v in errs.items()})
    assert max(errs.values()) < tol, errs
    flips = ((out[2][:, :100].cpu() > 0) != (ref["mask_preds"][:, :100] > 0)).float().mean().item()
    assert flips < (1e-3 if precision != "bf16" else 2e-2), flips
