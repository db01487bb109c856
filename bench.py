#!/usr/bin/env python
"""Headline benchmark (BASELINE.json): frames/s of the kernel-update + mask/depth forward
(`KernelUpdateIterHead.simple_test_mask_preds`) at 1024x2048, N=153, S=3.

    python bench.py --gpus N --steps K --warmup W

N > 1 runs one process per GPU over RCCL: either under an outer launcher (`python -m torch.distributed.run
--nproc-per-node N ... bench.py --gpus N`, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment) or, when no
WORLD_SIZE is set, by starting the other N-1 ranks itself (`self_launch`).

One "step" = one pass of the hot path over a batch of `--frames` synthetic frames per GPU, inputs
resident in HBM, the whole launch sequence replayed from a HIP graph.  Frames are independent, so
ranks share nothing on the data path ("weak" scaling, no collective inside the timed region).
Prints ONE JSON line on rank 0."""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

CFG5_FRAMES = 192       # frames per step of the cfg5 leg (four 48-frame parts)
WORKLOADS = {
    # BASELINE.json configs[1]: 1024x2048 -> stride-8 128x256, N = 100 + 53 (class defaults), S = 3
    "cfg2": dict(H=128, W=256, Nq=100, n_thing=80, n_stuff=53, S=3, F=2048),
    # configs[2]/[3]: the shipped video head (100 + 11 queries) at the same map size
    "cfg3": dict(H=128, W=256, Nq=100, n_thing=8, n_stuff=11, S=3, F=2048),
    # configs[4]: 1242x375 padded to 1248x384 -> 48x156 (HW = 7488 is not a multiple of 128), N = 200 + 53
    "cfg5": dict(H=48, W=156, Nq=200, n_thing=80, n_stuff=53, S=3, F=2048),
    # small variant for smoke runs
    "tiny": dict(H=16, W=32, Nq=100, n_thing=8, n_stuff=11, S=3, F=2048),
}


def stage_cfg(L, n_thing, n_stuff, F):
    return dict(type="KernelUpdateHead", num_thing_classes=n_thing, num_stuff_classes=n_stuff, num_classes=L,
                num_ffn_fcs=2, num_heads=8, num_cls_fcs=1, num_mask_fcs=1, feedforward_channels=F, in_channels=256,
                out_channels=256, dropout=0.0, mask_thr=0.5, conv_kernel_size=1, mask_upsample_stride=2,
                ffn_act_cfg=dict(type="ReLU", inplace=True), with_ffn=True,
                feat_transform_cfg=dict(conv_cfg=dict(type="Conv2d"), act_cfg=None),
                kernel_updator_cfg=dict(type="KernelUpdator", in_channels=256, feat_channels=256, out_channels=256,
                                        input_feat_shape=3, act_cfg=dict(type="ReLU", inplace=True),
                                        norm_cfg=dict(type="LN")),
                loss_cls=dict(type="FocalLoss", use_sigmoid=True), loss_mask=dict(type="CrossEntropyLoss", use_sigmoid=True),
                loss_dice=dict(type="DiceLoss"), loss_depth=dict(type="DepthLoss"), depth_act_mode="sigmoid")


def build_head(wl, precision, out_dtype, device, seed=0):
    from polyphonicformer_amd.registry import HEADS
    import polyphonicformer_amd.kernel_update  # noqa: F401
    import polyphonicformer_amd.kernel_update_head  # noqa: F401
    import polyphonicformer_amd.kernel_updator  # noqa: F401
    L = wl["n_thing"] + wl["n_stuff"]
    torch.manual_seed(seed)
    head = HEADS.build(dict(type="KernelUpdateIterHead", num_stages=wl["S"], assign_stages=wl["S"],
                            stage_loss_weights=[1] * wl["S"], num_proposals=wl["Nq"], num_thing_classes=wl["n_thing"],
                            num_stuff_classes=wl["n_stuff"], do_panoptic=True, merge_joint=True,
                            mask_head=stage_cfg(L, wl["n_thing"], wl["n_stuff"], wl["F"]),
                            test_cfg=dict(max_per_img=wl["Nq"])))
    head.init_weights()                       # the reference's init (xavier-uniform, kernel_update_head.py:193-205)
    head.eval().to(device)
    head.set_precision(precision, out_dtype)
    head.frame_invariant = False          # throughput legs: launch geometry tuned to the batch (the module API's default is True)
    return head


def synth_inputs(wl, B, seed, mask_bias=0.0):
    """BASELINE.md section 2, isolated-IterHead inputs (dense ~50 % foreground masks; SURVEY 8d's sparse variant is
    `--mask-bias -2`).  The kernels' work does not depend on the mask density (1-bit masks, dense MFMA)."""
    g = torch.Generator().manual_seed(seed)
    N = wl["Nq"] + wl["n_stuff"]
    H, W = wl["H"], wl["W"]
    return dict(x=torch.randn(B, 256, H, W, generator=g), dfe=torch.randn(B, 256, H, W, generator=g),
                k0=torch.randn(B, N, 256, generator=g), q0=torch.randn(1, 1, 256, generator=g).expand(B, N, 256),
                m0=torch.randn(B, N, H, W, generator=g) + mask_bias)


def time_op(fn, iters, warm=2):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def kernel_breakdown(plan, iters=4):
    """average duration (ms) of every launch class of one step, measured INSIDE the step: the plan's own launch sequence
    (binarize, S x (pool, query pre, query post, conv), x2 upsamples) runs `iters` times on the launch stream (torch's
    current stream = the stream the C ABI launches on) with a HIP event between consecutive launches, so every kernel
    sees the cache / DRAM state its predecessor leaves -- what a single-stream rocprofv3 trace of the step records
    (profiles/r03/e_single_stream_B24_kernel_stats.csv)."""
    from polyphonicformer_amd import engine as E
    p = plan
    seq = []                                    # (class name, launch)
    seq.append(("binarize", lambda: E.binarize(p.m0, out=p.bits)))
    k, q = p.k0, p.q0
    for s in range(p.S):
        last = s == p.S - 1
        o = p.stage_out[s]
        px = getattr(p, "poolx", False)
        if s > 0 and px:     # round 6: the x map was pooled by the previous stage's conv (ph_dynconv_poolx); depth_feats alone here
            seq.append(("pool_depth", lambda: E.pool_depth_only(p.dp, p.bits, p.N, p.HW, p.prec, p.partial_px, p.pcount_px)))
            part, cnt = p.partial_px, p.pcount_px
        else:
            seq.append(("pool", lambda: E.pool(p.xp, p.dp, p.bits, p.N, p.HW, p.prec, p.nsplit, out=p.partial, counts=p.pcount)))
            part, cnt = p.partial, p.pcount
        for name, ph in (("query_pre", 1), ("query_post", 2)):
            seq.append((name, lambda ph=ph, k=k, q=q, s=s, last=last, part=part, cnt=cnt: E.query_stage(
                part, p.bits, k, q, p.packs[s], p.N, p.HW, cls_sigmoid=last, outs=p.stage_out[s], workspace=p.ws,
                phases=ph | (_lib_wide() if getattr(p, "shares_gpu", False) else 0), kern_fmt=p.mode.kern_fmt, counts=cnt)))
        if not last and px:
            seq.append(("dynconv_poolx", lambda o=o: E.dynconv_poolx(p.xp, o["kern"], o["kbias"], p.N, p.HW, p.mode.conv, p.bits, p.partial_px)))
        elif not last:
            seq.append(("dynconv_bits", lambda o=o: E.dynconv(p.xp, o["kern"], o["kbias"], 0, p.N, p.HW, p.mode.conv, bits_out=p.bits)))
        elif getattr(p, "fused_up", False):      # final conv + x2 upsample in one kernel (the plan's own launches)
            seq.append(("dynconv_up2_mask", lambda o=o: E.dynconv_up2(p.xp, o["kern"], o["kbias"], 0, p.N, p.H, p.W, p.mode.conv, p.mask_up,
                                                                      logits_out=p.mask, out_dtype=p.out_code)))
            seq.append(("dynconv_up2_depth", lambda o=o: E.dynconv_up2(p.dp, o["kern"], o["kbias"], 1, p.N, p.H, p.W, p.mode.conv, p.depth_up,
                                                                       logits_out=p.depth if p.want_depth_lowres else None, out_dtype=p.out_code)))
        else:
            seq.append(("dynconv_logits", lambda o=o: E.dynconv(p.xp, o["kern"], o["kbias"], 0, p.N, p.HW, p.mode.conv,
                                                                logits_out=p.mask, out_dtype=p.out_code)))
            seq.append(("upsample2x", lambda: E.upsample2x(p.mask, out=p.mask_up)))       # the plan's order (engine.DecodePlan.stages)
            seq.append(("dynconv_logits", lambda o=o: E.dynconv(p.dp, o["kern"], o["kbias"], 1, p.N, p.HW, p.mode.conv,
                                                                logits_out=p.depth, out_dtype=p.out_code)))
            seq.append(("upsample2x", lambda: E.upsample2x(p.depth, out=p.depth_up)))
        k, q = o["obj"], o["dobj"]
    for _, fn in seq:                           # one untimed pass
        fn()
    tot, cnt = {}, {}
    for _ in range(iters):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(seq) + 1)]
        torch.cuda.synchronize()
        ev[0].record()
        for i, (_, fn) in enumerate(seq):
            fn()
            ev[i + 1].record()
        torch.cuda.synchronize()
        for i, (name, _) in enumerate(seq):
            tot[name] = tot.get(name, 0.0) + ev[i].elapsed_time(ev[i + 1])
            cnt[name] = cnt.get(name, 0) + 1
    t = {name: tot[name] / cnt[name] for name in tot}
    t["ingest"] = time_op(lambda: E.ingest(p.x, p.prec, out=p.xp), 4) if getattr(p, "x", None) is not None else 0.0
    npx = p.S - 1 if getattr(p, "poolx", False) else 0
    counts = dict(ingest=0 if getattr(p, "feat_is_bf16", False) else 2, binarize=1, pool=p.S - npx, pool_depth=npx, query_pre=p.S, query_post=p.S,
                  dynconv_bits=p.S - 1 - npx, dynconv_poolx=npx, dynconv_logits=2, upsample2x=2, dynconv_up2_mask=1, dynconv_up2_depth=1)
    counts = {k: v for k, v in counts.items() if k in t}
    return t, counts


def _lib_wide():
    from polyphonicformer_amd import _lib
    return _lib.PH_QUERY_WIDE


def algorithmic_bytes(plan, kernel):
    """ALGORITHMIC bytes one launch of `kernel` must move."""
    p = plan
    P = p.mode.FP
    from polyphonicformer_amd.engine import hw_padded, n_padded
    HWp, Npad = hw_padded(p.HW), n_padded(p.N)
    eo = 4 if p.out_dtype == torch.float32 else 2
    feat = p.B * 256 * p.HW * 2 * P            # one feature map, all planes
    bits = p.B * p.N * p.HW // 8
    if kernel == "pool":
        return 2 * feat + bits
    if kernel in ("dynconv_bits", "dynconv_poolx", "pool_depth"):      # one plane in, the mask bits out / in (+ the pooled sums: KBs)
        return feat + bits
    if kernel == "dynconv_logits":
        return feat + p.B * p.N * p.HW * eo
    if kernel == "upsample2x":
        return p.B * p.N * p.HW * eo * 5
    if kernel == "dynconv_up2_mask":        # plane in, low-resolution logits + the x2 upsampled logits out
        return feat + p.B * p.N * p.HW * eo * 5
    if kernel == "dynconv_up2_depth":       # plane in, the x2 upsampled logits out
        return feat + p.B * p.N * p.HW * eo * (5 if p.want_depth_lowres else 4)
    if kernel == "ingest":
        return p.B * 256 * p.HW * 4 + feat
    if kernel == "binarize":
        return p.B * p.N * p.HW * p.m0.element_size() + bits
    return None


def roofline_of(kplan, times, counts, nplans):
    """roofline object of the launch class with the largest share of a step's GPU time.  The fused final stage's two instantiations
    (mask form with the low-resolution logits, depth form without) are ONE class, `dynconv_up2`: bytes and time of both launches"""
    per_step = {k: times[k] * counts.get(k, 0) * nplans for k in times if algorithmic_bytes(kplan, k) and k != "ingest"}
    klass = dict(per_step)
    if "dynconv_up2_mask" in klass:
        klass["dynconv_up2"] = klass.pop("dynconv_up2_mask") + klass.pop("dynconv_up2_depth")
    dom = max(klass, key=lambda k: klass[k])
    if dom == "dynconv_up2":
        ab = algorithmic_bytes(kplan, "dynconv_up2_mask") + algorithmic_bytes(kplan, "dynconv_up2_depth")
        t_ms, launches = times["dynconv_up2_mask"] + times["dynconv_up2_depth"], 2
    else:
        ab, t_ms, launches = algorithmic_bytes(kplan, dom) * 1, times[dom], 1
    achieved = ab / (t_ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 4),
            "algorithmic_bytes_per_launch": ab // launches, "avg_launch_ms": round(t_ms / launches, 4), "launches_in_the_figure": launches,
            "frames_per_launch": kplan.B, "share_of_hbm_bound_launch_time": round(klass[dom] / sum(klass.values()), 4)}


def algorithmic_rates(wl, N, frames_per_launch, fps_per_gpu, precision):
    """B_alg = (S+1)*2*C*HW*e + N*HW*e (initial mask logits) + 2*N*HW*e + 2*N*4*HW*e + S*P*e_w / frames,
    F_alg = S*8*N*C*HW + S*N*(7.73e6 + 2048 N + 512 L); e = 2 (bf16 planes and outputs), weights bf16"""
    HW, S, C = wl["H"] * wl["W"], wl["S"], 256
    L = wl["n_thing"] + wl["n_stuff"]
    e = 2
    pw = 4.02e6 * 2 * (2 if precision != "bf16" else 1)
    b_alg = (S + 1) * 2 * C * HW * e + N * HW * e + 2 * N * HW * e + 2 * N * 4 * HW * e + S * pw / frames_per_launch
    f_alg = S * 8 * N * C * HW + S * N * (7.73e6 + 2048 * N + 512 * L)
    gbps, tf = b_alg * fps_per_gpu / 1e9, f_alg * fps_per_gpu / 1e12
    return {"bytes_per_frame": int(b_alg), "flop_per_frame": int(f_alg), "achieved_GBps": round(gbps, 1),
            "fraction_hbm": round(gbps / 8000.0, 4), "achieved_TFLOPs": round(tf, 1), "fraction_mfma_bf16": round(tf / 2500.0, 4)}


def mode_leg(wl, mode, dev, B, parts, fp32_inputs=False, steps=10, breakdown=False, mask_bias=0.0):
    """frames/s of simple_test_mask_preds in precision mode `mode` (engine.MODES): B frames as `parts` part-batches on
    skewed streams from ONE HIP graph, features resident in the mode's own plane format (or fp32 NCHW + ingest).
    `breakdown`: also the per-launch timings of one part's sequence, the roofline of its dominant launch class and the whole-path
    algorithmic rates (the default run's cfg5 leg)"""
    from polyphonicformer_amd.engine import DualDecodePlan, MODES
    out_dtype = {"bf16": torch.bfloat16, "mixed": torch.float16, "mixed16": torch.float16, "fp16": torch.float16, "fp32": torch.float32}[mode]
    head = build_head(wl, mode, out_dtype, dev)
    N = wl["Nq"] + wl["n_stuff"]
    plan = head._plan(B // parts, N, wl["H"], wl["W"], dev)
    runner = DualDecodePlan(plan.packs, B, N, wl["H"], wl["W"], plan.mode, out_dtype, dev, parts=parts)
    inp = synth_inputs(wl, B, seed=99, mask_bias=mask_bias)
    gin = [inp[k].to(dev) for k in ("x", "dfe", "k0", "q0", "m0")]
    fdt = MODES[mode].feat_dtype
    if fdt is not None and not fp32_inputs:
        gin[0], gin[1] = gin[0].to(fdt), gin[1].to(fdt)
        if out_dtype != torch.float32:
            gin[4] = gin[4].to(out_dtype)            # 16-bit mask logits with 16-bit features (see main)
    runner.set_inputs(*gin)
    runner.capture()
    t = time_op(runner.replay, steps, warm=4)
    out = {"value": round(B / (t * 1e-3), 2), "unit": "frames/s", "ms_per_step": round(t, 4), "frames_per_step": B, "streams": parts,
           "feature_input_dtype": "fp32 (ingest inside the step)" if (fp32_inputs or fdt is None) else str(fdt), "output_dtype": str(out_dtype),
           "workload": f"{wl['H'] * 8}x{wl['W'] * 8}, N={N}, S={wl['S']}"}
    if breakdown:
        kplan = runner.halves[0]
        times, counts = kernel_breakdown(kplan)
        out["kernels_ms"] = {k: round(v, 4) for k, v in times.items()}
        out["roofline"] = roofline_of(kplan, times, counts, parts)
        out["algorithmic"] = algorithmic_rates(wl, N, kplan.B, B / (t * 1e-3), mode)
        # the query side at this shape: weights streamed per workgroup, 1.25 GF per stage and frame on MFMA
        q = (times["query_pre"] + times["query_post"]) * kplan.S
        hb = sum(times[k] * counts.get(k, 0) for k in times if algorithmic_bytes(kplan, k) and k != "ingest")
        out["query_share_of_one_part"] = round(q / (q + hb), 4)
    del runner, plan, head, gin, inp
    torch.cuda.empty_cache()
    return out


def kernel_head_leg(wl, head, precision, out_dtype, dev, B=16, steps=10):
    """secondary number (reported next to the headline): KernelHead post-neck (a1) + the S-stage decode (a6)
    with the bf16-plane / mask-bit hand-off (no ingest pass), one stream, HIP graph."""
    from polyphonicformer_amd.registry import HEADS
    from polyphonicformer_amd import engine as E
    import polyphonicformer_amd.kernel_head  # noqa: F401
    L = wl["n_thing"] + wl["n_stuff"]
    torch.manual_seed(1)
    kh = HEADS.build(dict(type="KernelHead", num_proposals=wl["Nq"], num_classes=L, num_thing_classes=wl["n_thing"],
                          num_stuff_classes=wl["n_stuff"], cat_stuff_mask=True, feat_downsample_stride=2, feat_refine=False,
                          use_binary=True, proposal_feats_with_obj=True, kernel_init_std=1, conv_normal_init=True,
                          loss_seg=dict(type="FocalLoss", use_sigmoid=True)))
    kh.init_weights()
    kh.eval().to(dev)
    kh.set_precision(precision)
    kh.emit_fp32_features = False
    N, H, W = wl["Nq"] + wl["n_stuff"], wl["H"], wl["W"]
    kplan = E.KernelHeadPlan(kh._get_pack(dev), B, H, W, wl["n_thing"], L, True, dev, want_f32=False)
    g = torch.Generator().manual_seed(3)
    kplan.set_inputs([torch.randn(B, 256, H, W, generator=g).relu().to(dev) for _ in range(3)])
    dplan = E.DecodePlan([h.stage_pack(dev, precision) for h in head.mask_head], B, N, H, W, E.MODES[precision], out_dtype, dev)
    q0 = kh._get_pack(dev).w_dd_f32.reshape(1, 1, 256).expand(B, N, 256)

    def run():
        kplan.run()
        dplan.run_from_planes(kplan.xp, kplan.dp, kplan.bits, kplan.proposal, q0)

    def run_a1():
        kplan.run()

    run()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        run()
    t_all = time_op(graph.replay, steps)
    t_a1 = time_op(run_a1, steps)
    a1_timeouts = kplan.timeouts()
    # the same with fp16 mask / seg / depth logits out of KernelHead (the decode consumes the mask BITS; the logits are API
    # outputs of simple_test_rpn): 19 MB per frame less
    t_a1_h = None
    if kplan.onepass:
        try:
            kplan_h = E.KernelHeadPlan(kh._get_pack(dev), B, H, W, wl["n_thing"], L, True, dev, want_f32=False, logit_dtype=torch.float16)
            kplan_h.set_inputs(kplan.f)
            t_a1_h = time_op(kplan_h.run, steps)
            del kplan_h
        except Exception as e:
            t_a1_h = repr(e)
    # a1 as the WHOLE head runs it: the three maps arrive as the neck's 16-bit planes ([1][B][256][HWp], what
    # SemanticFPNWrapper.forward_planes hands over -- no fp32 round trip between neck and KernelHead) and the logits leave as fp16;
    # a1 + a6 from one graph in that form
    handoff = None
    if kplan.onepass and precision in ("fp16", "bf16") and E.hw_padded(H * W) == H * W:
        try:
            fdt = torch.float16 if precision == "fp16" else torch.bfloat16
            kplan_p = E.KernelHeadPlan(kh._get_pack(dev), B, H, W, wl["n_thing"], L, True, dev, want_f32=False, logit_dtype=torch.float16)
            kplan_p.set_inputs([f.to(fdt).view(torch.int16).reshape(1, B, 256, H * W).contiguous() for f in kplan.f])

            def run_p():
                kplan_p.run()
                dplan.run_from_planes(kplan_p.xp, kplan_p.dp, kplan_p.bits, kplan_p.proposal, q0)

            run_p()
            torch.cuda.synchronize()
            gp = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gp):
                run_p()
            t_p = time_op(gp.replay, steps)
            t_a1_p = time_op(kplan_p.run, steps)
            a1b_p = 3 * 256 * H * W * 2 + 2 * 256 * H * W * 2 + (N + L + 1) * H * W * 2 + N * H * W // 8 + 256 * H * W * 2
            a6b_p = algorithmic_rates(wl, N, B, 1.0, precision)["bytes_per_frame"] - N * H * W * 2
            handoff = {"a1_only_ms_per_step": round(t_a1_p, 4), "a1_plus_a6_ms_per_step": round(t_p, 4),
                       "a1_plus_a6_frames_per_s": round(B / (t_p * 1e-3), 1), "a1_alg_bytes_per_frame": a1b_p,
                       "a1_plus_a6_fraction_hbm": round((a1b_p + a6b_p) * (B / (t_p * 1e-3)) / 8e12, 4),
                       "note": "16-bit planes in (the neck's hand-off), fp16 logits out; a1 bytes = 3 planes read + x / dfe planes + fp16 logits + bits "
                               "+ the pooling's read of x"}
            del kplan_p, gp
        except Exception as e:
            handoff = {"error": repr(e)}
    # the same on two streams, a second half-batch of B frames one phase behind the first (as the headline does for a6)
    two = None
    try:
        kplan2 = E.KernelHeadPlan(kh._get_pack(dev), B, H, W, wl["n_thing"], L, True, dev, want_f32=False)
        kplan2.set_inputs([torch.randn(B, 256, H, W, generator=g).relu().to(dev) for _ in range(3)])
        dplan2 = E.DecodePlan([h.stage_pack(dev, precision) for h in head.mask_head], B, N, H, W, E.MODES[precision], out_dtype, dev)
        sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

        def issue():
            cur = torch.cuda.current_stream()
            sa.wait_stream(cur)
            sb.wait_stream(cur)
            with torch.cuda.stream(sa):
                kplan.run()
                skew = torch.cuda.Event()
                skew.record(sa)
                dplan.run_from_planes(kplan.xp, kplan.dp, kplan.bits, kplan.proposal, q0)
            with torch.cuda.stream(sb):
                sb.wait_event(skew)
                kplan2.run()
                dplan2.run_from_planes(kplan2.xp, kplan2.dp, kplan2.bits, kplan2.proposal, q0)
            cur.wait_stream(sa)
            cur.wait_stream(sb)

        issue()
        torch.cuda.synchronize()
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2):
            issue()
        t2 = time_op(g2.replay, steps)
        two = {"frames_per_step": 2 * B, "frames_per_s": round(2 * B / (t2 * 1e-3), 1), "ms_per_step": round(t2, 4)}
    except Exception as e:
        two = {"error": repr(e)}
    # the same at the headline's batch: a1 over 128 frames in ONE persistent launch (the frames take turns on the CUs), then a6
    # as four 32-frame parts on four streams, each a phase behind the previous -- the a6 form the headline times
    big = None
    try:
        del kplan2, dplan2
    except NameError:
        pass
    torch.cuda.empty_cache()
    try:
        B96, parts = 128, 4          # the headline's step: 4 parts of 32 frames (96 = 4 x 24 until round 6)
        kp = E.KernelHeadPlan(kh._get_pack(dev), B96, H, W, wl["n_thing"], L, True, dev, want_f32=False)
        kp.set_inputs([torch.randn(B96, 256, H, W, generator=g).relu().to(dev) for _ in range(3)])
        packs = [h.stage_pack(dev, precision) for h in head.mask_head]
        dps = [E.DecodePlan(packs, B96 // parts, N, H, W, E.MODES[precision], out_dtype, dev) for _ in range(parts)]
        for d_ in dps:
            d_.shares_gpu = True
        q96 = kh._get_pack(dev).w_dd_f32.reshape(1, 1, 256).expand(B96 // parts, N, 256)
        sts = [torch.cuda.Stream() for _ in range(parts)]

        def issue96():
            cur = torch.cuda.current_stream()
            kp.run()
            prev = torch.cuda.Event()
            prev.record(cur)
            n = B96 // parts
            for i, (d_, st) in enumerate(zip(dps, sts)):
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    st.wait_event(prev)
                    ev = torch.cuda.Event()
                    d_.on_first_pool = lambda ev=ev, st=st: ev.record(st)
                    d_.run_from_planes(kp.xp[:, i * n:(i + 1) * n], kp.dp[:, i * n:(i + 1) * n], kp.bits[i * n:(i + 1) * n],
                                       kp.proposal[i * n:(i + 1) * n], q96)
                    prev = ev
            for st in sts:
                cur.wait_stream(st)

        issue96()
        torch.cuda.synchronize()
        g96 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g96):
            issue96()
        t96 = time_op(g96.replay, max(3, steps // 2))
        a6b = algorithmic_rates(wl, N, B96 // parts, 1.0, precision)["bytes_per_frame"] - N * H * W * 2
        a1b = 3 * 256 * H * W * 4 + 2 * 256 * H * W * 2 + (N + L + 1) * H * W * 4 + N * H * W // 8 + 256 * H * W * 2
        big = {"frames_per_step": B96, "frames_per_s": round(B96 / (t96 * 1e-3), 1), "ms_per_step": round(t96, 4),
               "fraction_hbm": round((a1b + a6b) * (B96 / (t96 * 1e-3)) / 8e12, 4), "a1_onepass_timeouts": kp.timeouts(),
               "note": f"a1 over {B96} frames in one persistent launch, then a6 as four {B96 // parts}-frame parts on four streams (the headline's a6 form; 96 = 4 x 24 frames until round 6)"}
        del kp, dps, g96
    except Exception as e:
        big = {"error": repr(e)}
    torch.cuda.empty_cache()
    return {"frames_per_step": B, "a1_plus_a6_frames_per_s": round(B / (t_all * 1e-3), 1), "a1_plus_a6_ms_per_step": round(t_all, 4),
            "headline_batch_four_streams": big, "neck_handoff_inputs": handoff,
            "a1_only_ms_per_step": round(t_a1, 4), "a1_onepass_timeouts": a1_timeouts, "a1_form": "one-pass (ph_khead_onepass)" if kplan.onepass else "two-pass (ph_khead_fused)",
            "a1_only_ms_per_step_fp16_logits": round(t_a1_h, 4) if isinstance(t_a1_h, float) else t_a1_h, "two_streams": two,
            "a1_alg_bytes_per_frame": int(3 * 256 * H * W * 4 + 2 * 256 * H * W * 2 + (N + L + 1) * H * W * 4 + N * H * W // 8 + 256 * H * W * 2),
            "a1_plus_a6_fraction_hbm": round(((3 * 256 * H * W * 4 + 2 * 256 * H * W * 2 + (N + L + 1) * H * W * 4 + N * H * W // 8 + 256 * H * W * 2)
                                              + algorithmic_rates(wl, N, B, 1.0, precision)["bytes_per_frame"] - N * H * W * 2) * (B / (t_all * 1e-3)) / 8e12, 4),
            "note": "KernelHead post-neck (3 x conv1x1+GN+ReLU in ONE read of the maps: persistent kernel, GroupNorm sums exchanged between "
                    "the workgroups inside the launch; the static 1x1 convs, x = sem + loc and the mask bits in the same launch; object "
                    "pooling) + 3-stage decode, 16-bit plane + mask-bit hand-off, fp32 post-neck inputs resident in HBM.  "
                    "a1_alg_bytes_per_frame = maps read once + x / dfe planes + fp32 logits + bits + the pooling's read of x; "
                    "a1_plus_a6_fraction_hbm = (that + SURVEY 8d's B_alg less the initial logit read) x frames/s / 8 TB/s"}


def neck_leg(wl, precision, dev, B=16, steps=5):
    """secondary number: SemanticFPNWrapper.forward, the step that produces the hot path's three input
    maps, at the cfg2 FPN sizes (strides 4..32 of 1024x2048), fp32 NCHW levels resident in HBM"""
    from polyphonicformer_amd.registry import NECKS
    import polyphonicformer_amd.semantic_fpn  # noqa: F401
    torch.manual_seed(2)
    m = NECKS.build(dict(type="SemanticFPNWrapper", in_channels=256, feat_channels=256, out_channels=256, start_level=0,
                         end_level=3, upsample_times=2, positional_encoding=dict(type="SinePositionalEncoding", num_feats=128, normalize=True),
                         cat_coors=False, cat_coors_level=3, fuse_by_cat=False, return_list=False, num_aux_convs=2,
                         norm_cfg=dict(type="GN", num_groups=32, requires_grad=True)))
    m.init_weights()
    m.eval().to(dev)
    m.set_precision(precision)
    H0, W0 = wl["H"] * 2, wl["W"] * 2
    g = torch.Generator().manual_seed(4)
    feats = [torch.randn(B, 256, H0 >> i, W0 >> i, generator=g).to(dev) for i in range(4)]
    t = time_op(lambda: m(feats), steps)
    flop = 2 * 256 * 256 * (9 * (4 * wl["H"] * wl["W"] + 2 * wl["H"] * wl["W"] // 4 + wl["H"] * wl["W"] // 16) + 3 * wl["H"] * wl["W"])
    return {"frames_per_step": B, "frames_per_s": round(B / (t * 1e-3), 1), "ms_per_step": round(t, 4),
            "mfma_TFLOPs": round(flop * B / (t * 1e-3) / 1e12, 1), "flop_per_frame": flop,
            "note": "7 conv3x3 + GN + ReLU towers, x2 upsamples, level sum, conv_pred + 2 aux convs; channels-last implicit GEMM"}


def assign_leg(dev, B=16, N=100, G=40, H=128, W=256, steps=20):
    """secondary number (first part): the pixel sums of the mask Hungarian assigner's costs for a batch of
    training crops (512 x 1024 at assign stride 4), one `ph_match_sums` launch; the reference does three einsums + four
    row sums per image (polyphonic/funcs/assigner.py:113-129,178-194).  Also the host part per image (cost algebra +
    scipy Hungarian) and the same pixel sums by the CPU oracle."""
    from polyphonicformer_amd import assigner as A
    g = torch.Generator().manual_seed(5)
    z = (torch.randn(B, N, H, W, generator=g) * 2).to(dev)
    t = (torch.rand(B, G, H, W, generator=g) > 0.7).float().to(dev)
    v = (torch.rand(B, H, W, generator=g) > 0.1).float().to(dev)
    lib = A._lib.load()
    part = torch.empty((B, lib.ph_match_nsplit(H * W, B), lib.ph_match_record_floats(N, G)), dtype=torch.float32, device=dev)

    def kernel():
        A._lib.check(lib.ph_match_sums(A._lib.ptr(z), A._lib.ptr(t), A._lib.ptr(v), A._lib.ptr(part), B, N, G, H * W,
                                       A._lib.stream_ptr()), "ph_match_sums")

    ms = time_op(kernel, steps)
    nbytes = B * (N + G + 1) * H * W * 4
    a = A.build_assigner(dict(type='MaskHungarianAssignerWithDepth', cls_cost=dict(type='FocalLossCost', weight=2.0),
                              dice_cost=dict(type='DiceCost', weight=4.0, pred_act=True),
                              mask_cost=dict(type='MaskCost', weight=1.0, pred_act=True)))
    cls, lab = torch.randn(N, 8, generator=g).to(dev), torch.randint(0, 8, (G,), generator=g).to(dev)
    a.assign(z[0], cls, t[0], lab, None, gt_valid=v[0])          # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(B):
        a.assign(z[i], cls, t[i], lab, None, gt_valid=v[i])
    torch.cuda.synchronize()
    per_img = (time.perf_counter() - t0) / B * 1e3
    out = {"images_per_launch": B, "N": N, "G": G, "HxW": [H, W], "pixel_sums_ms_per_launch": round(ms, 4),
           "pixel_sums_GBps": round(nbytes / ms / 1e6, 1), "bytes_per_launch": nbytes,
           "assign_ms_per_image_incl_host_hungarian": round(per_img, 3),
           "note": "one pass over mask logits + gt masks + valid (fp32 in, sigmoid fused, bf16 hi/lo MFMA over the pixel axis)"}
    return out


def full_head_leg(wl, head, precision, dev, B=16, steps=5):
    """secondary number: the whole head as `Polyphonic.simple_test` wires it (polyphonic_former.py:145-161), through the
    module API: FPN levels -> rpn_head.simple_test_rpn (SemanticFPNWrapper + KernelHead post-neck) ->
    roi_head.simple_test_mask_preds (3 stages + upsample); fp32 NCHW tensors at every API boundary"""
    from polyphonicformer_amd.registry import HEADS
    import polyphonicformer_amd.kernel_head  # noqa: F401
    L = wl["n_thing"] + wl["n_stuff"]
    torch.manual_seed(5)
    neck = dict(type="SemanticFPNWrapper", in_channels=256, feat_channels=256, out_channels=256, start_level=0, end_level=3,
                upsample_times=2, positional_encoding=dict(type="SinePositionalEncoding", num_feats=128, normalize=True),
                cat_coors=False, cat_coors_level=3, fuse_by_cat=False, return_list=False, num_aux_convs=2,
                norm_cfg=dict(type="GN", num_groups=32, requires_grad=True))
    kh = HEADS.build(dict(type="KernelHead", num_proposals=wl["Nq"], num_classes=L, num_thing_classes=wl["n_thing"],
                          num_stuff_classes=wl["n_stuff"], cat_stuff_mask=True, feat_downsample_stride=2, feat_refine=False,
                          use_binary=True, proposal_feats_with_obj=True, kernel_init_std=1, conv_normal_init=True,
                          loss_seg=dict(type="FocalLoss", use_sigmoid=True), localization_fpn=neck))
    kh.init_weights()
    kh.eval().to(dev)
    kh.set_precision(precision)
    kh.frame_invariant = False            # throughput leg (the video legs keep the module API's frame-invariant default)
    H0, W0 = wl["H"] * 2, wl["W"] * 2
    g = torch.Generator().manual_seed(6)
    feats = tuple(torch.randn(B, 256, H0 >> i, W0 >> i, generator=g).to(dev) for i in range(4))
    metas = [dict(img_shape=(wl["H"] * 8, wl["W"] * 8, 3), ori_shape=(wl["H"] * 8, wl["W"] * 8, 3),
                  batch_input_shape=(wl["H"] * 8, wl["W"] * 8))] * B

    def run():
        (pf, xf, mp, cs, seg, df, dp, dpr, aspp) = kh.simple_test_rpn(feats, metas)
        return head.simple_test_mask_preds(xf, pf, mp, cs, metas, depth_preds=dpr, depth_feats=df, depth_proposal=dp)

    t = time_op(run, steps)
    out = {"frames_per_step": B, "frames_per_s": round(B / (t * 1e-3), 1), "ms_per_step": round(t, 4),
           "note": "FPN levels (fp32 NCHW) -> SemanticFPNWrapper -> KernelHead -> KernelUpdateIterHead.simple_test_mask_preds, "
                   "module API (fresh output tensors per call), eager launches"}
    try:        # the same module-API call sequence captured once into a HIP graph and replayed
        run()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            keep = run()
        tg = time_op(graph.replay, steps)
        out["hip_graph"] = {"frames_per_s": round(B / (tg * 1e-3), 1), "ms_per_step": round(tg, 4)}
        del keep, graph
    except Exception as e:
        out["hip_graph"] = {"error": repr(e)}
    try:        # two module pipelines on two streams, the second one a phase behind: its neck (matrix-pipe bound) runs under
        #         the first one's KernelHead + decode (HBM bound) -- what a serving loop with two frame batches in flight does
        import copy
        # a second pair of modules with the same weights and its own plans: the originals' plans (GBs of device buffers, HIP
        # streams and events) are taken out while the modules are copied
        neck_ = getattr(kh, "localization_fpn", None)
        if neck_ is not None:
            neck_.tower_streams = False     # two pipelines x four tower streams in ONE captured graph: hipStreamEndCapture segfaults
        held = [(m_, m_._plans) for m_ in (kh, head, neck_) if m_ is not None and hasattr(m_, "_plans")]
        for m_, _ in held:
            m_._plans = {}
        try:
            kh2, head2 = copy.deepcopy(kh), copy.deepcopy(head)
        finally:
            for m_, pl in held:
                m_._plans = {} if m_ is neck_ else pl        # the neck re-plans without tower streams for this leg
        kh2._pack = None
        feats2 = tuple(f.clone() for f in feats)
        sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

        def one(k_, h_, f_):
            (pf, xf, mp, cs, seg, df, dp, dpr, aspp) = k_.simple_test_rpn(f_, metas)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            return ev, h_.simple_test_mask_preds(xf, pf, mp, cs, metas, depth_preds=dpr, depth_feats=df, depth_proposal=dp)

        def issue():
            cur = torch.cuda.current_stream()
            sa.wait_stream(cur)
            sb.wait_stream(cur)
            with torch.cuda.stream(sa):
                skew, ra = one(kh, head, feats)
            with torch.cuda.stream(sb):
                sb.wait_event(skew)
                _, rb = one(kh2, head2, feats2)
            cur.wait_stream(sa)
            cur.wait_stream(sb)
            return ra, rb

        issue()
        torch.cuda.synchronize()
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2):
            keep2 = issue()
        t2 = time_op(g2.replay, steps)
        out["two_streams"] = {"frames_per_step": 2 * B, "frames_per_s": round(2 * B / (t2 * 1e-3), 1), "ms_per_step": round(t2, 4)}
        del keep2, g2, kh2, head2
    except Exception as e:
        out["two_streams"] = {"error": repr(e)}
    neck_ = getattr(kh, "localization_fpn", None)
    if neck_ is not None and getattr(neck_, "tower_streams", True) is False:
        neck_.tower_streams = True
        neck_._plans = {}
    return out


def _video_pipeline(dev, precision):
    """the shipped video head (poly_r50_cityscapes_1x: 100 + 11 queries) with this package's neck, random-init weights;
    classification biases raised so that an un-trained network yields thing segments"""
    from polyphonicformer_amd.registry import HEADS, ConfigDict
    from polyphonicformer_amd import video as V
    import polyphonicformer_amd.kernel_head, polyphonicformer_amd.track_head  # noqa: F401,E401
    wl = WORKLOADS["cfg3"]
    L = wl["n_thing"] + wl["n_stuff"]
    torch.manual_seed(7)
    neck = dict(type="SemanticFPNWrapper", in_channels=256, feat_channels=256, out_channels=256, start_level=0, end_level=3,
                upsample_times=2, positional_encoding=dict(type="SinePositionalEncoding", num_feats=128, normalize=True),
                cat_coors=False, cat_coors_level=3, fuse_by_cat=False, return_list=False, num_aux_convs=2,
                norm_cfg=dict(type="GN", num_groups=32, requires_grad=True))
    kh = HEADS.build(dict(type="KernelHead", num_proposals=wl["Nq"], num_classes=L, num_thing_classes=wl["n_thing"],
                          num_stuff_classes=wl["n_stuff"], cat_stuff_mask=True, feat_downsample_stride=2, feat_refine=False,
                          use_binary=True, proposal_feats_with_obj=True, kernel_init_std=1, conv_normal_init=True,
                          loss_seg=dict(type="FocalLoss", use_sigmoid=True), localization_fpn=neck))
    kh.init_weights()
    kh.eval().to(dev)
    kh.set_precision(precision)
    ih = build_head(wl, precision, torch.float32, dev, seed=3)
    ih.frame_invariant = True             # video: a clip's frames through one launch equal the per-frame loop bit for bit
    ih.test_cfg = ConfigDict(max_per_img=wl["Nq"], mask_thr=0.5, merge_stuff_thing=dict(overlap_thr=0.0, instance_score_thr=0.3))
    with torch.no_grad():      # un-trained masks overlap heavily: accept every segment that wins pixels (overlap_thr 0) ...
        ih.mask_head[-1].fc_cls.bias.fill_(1.0)      # ... and let every query pass the score threshold (sigmoid(1) = 0.73)
    th = HEADS.build(dict(type="QuasiDenseMaskEmbedHeadGTMask", norm_cfg=dict(type="GN", num_groups=32)))
    th.init_weights()
    th.to(dev).eval()
    th.precision = precision
    cfg = dict(init_score_thr=0.35, obj_score_thr=0.3, match_score_thr=0.5, memo_tracklet_frames=5, memo_backdrop_frames=1,
               memo_momentum=0.8, nms_conf_thr=0.5, nms_backdrop_iou_thr=0.3, nms_class_iou_thr=0.7, with_cats=True,
               match_metric="bisoftmax")
    return V.VideoFramePipeline(kh, ih, th, cfg), cfg, wl


def _video_frame(base, f, period, _cache={}):
    """synthetic FPN levels of global frame f of the video: the base levels rolled by a frame-dependent offset.  The `period` distinct
    frames are made ONCE and stay resident (round 6: torch.roll inside the timed loops was 1.4 ms of GPU time per 8-frame step of the
    cfg4 leg -- the bench's harness, not the path: a real stream's FPN levels come from the backbone and are already in HBM)"""
    key = (tuple(t.data_ptr() for t in base), f % period, period)
    if key not in _cache:
        if len(_cache) > 64:
            _cache.clear()
        _cache[key] = (tuple(torch.roll(t, (f % period, 2 * (f % period)), dims=(2, 3)) for t in base), base)    # `base` kept alive: its pointers are the key
    return _cache[key][0]


def cfg4_run(dev, world, rank, backend, precision="fp16", clip_frames=2, steps=4, warmup=1, collect_ids=False):
    """BASELINE configs[3]: clips sharded one per GPU, the track records all-gathered on RCCL, the tracker replayed in frame
    order.  The video is `world * clip_frames` frames per step; rank r owns the contiguous clip r (`dist.shard_frames`,
    SURVEY 8e).  One step per rank = polyphonic/apis/video_inference.py:8-31's loop body for its frames up to the tracker:
    PolyphonicVideo.simple_test after extract_feat (polyphonic_former_video.py:327-389: neck, both heads, panoptic merge,
    things -> boxes -> FPN RoIAlign -> track embeddings) with `records_only`, then ONE `dist.allgather_track_records` and
    `video.replay_tracking` of all frames of the step in frame order with the stream's persistent tracker (:391-402).
    Returns timings (max over ranks) and, with collect_ids, the track ids of every frame."""
    import torch.distributed as dist
    from polyphonicformer_amd import dist as D, video as V
    assert dist.is_initialized() and dist.get_world_size() == world, "process group does not span --gpus ranks"
    pipe, tcfg, wl = _video_pipeline(dev, precision)
    H8, W8 = wl["H"] * 8, wl["W"] * 8
    g = torch.Generator().manual_seed(31)               # the same video on every rank; each takes its own frames
    base = [torch.randn(1, 256, H8 // s, W8 // s, generator=g).to(dev) for s in (4, 8, 16, 32)]
    meta = [dict(img_shape=(H8, W8, 3), ori_shape=(H8, W8, 3), batch_input_shape=(H8, W8))]
    cdev = dev if backend == "nccl" else torch.device("cpu")
    tracker = V.QuasiDenseEmbedTracker(**tcfg)
    per_step = world * clip_frames
    ids_log, t_heads, t_coll, t_replay, cnt, seen = {}, [], [], [], 1, []
    # round 4: the rank's frames go through video.VideoStreamRunner (heads from ONE HIP graph, merge + record on the device);
    # PH_VIDEO_EAGER=1 restores the module-API call per frame (same records, same ids)
    runner = None if os.environ.get("PH_VIDEO_EAGER") else V.VideoStreamRunner(pipe, meta[0])

    def frames_of(step):
        return [step * per_step + f for f in D.shard_frames(per_step, rank, world)]

    def begin(step):
        if runner is not None:                  # the clip's heads start (HIP graph replays on the slots' streams); no wait
            # round 6: the frames are BORROWED -- this loop leaves a clip's tensors alone until its records are back, as the reference's
            # loop does (the backbone's outputs live until the frame is done) -- so no staging copy of the levels into the graph's
            # static inputs (356 MB of traffic per frame); PH_CFG4_BORROW=0 times the copying form
            runner.records_begin([_video_frame(base, f, 6) for f in frames_of(step)], borrowed=os.environ.get("PH_CFG4_BORROW", "1") != "0")

    def finish(step):
        """the records of this rank's frames of `step` (the synchronising half), packed for the all-gather"""
        mine = frames_of(step)
        if runner is not None:
            outs = runner.records_end()
        else:
            outs = [pipe.simple_test(_video_frame(base, f, 6), meta, records_only=True) for f in mine]
        recs, cnts = [], []
        for seg_ids, rec in outs:
            if rec is None:
                rec = (torch.zeros(0, 5), torch.zeros(0, dtype=torch.int64), torch.zeros(0, 256, device=dev))
            r, n = D.pack_track_records(*[t.to(cdev) for t in rec])
            recs.append(r)
            cnts.append(n)
        return mine, recs, cnts

    def gather_and_replay(mine, recs, cnts):
        nonlocal cnt
        t1 = time.perf_counter()
        allrec = D.allgather_track_records(mine, recs, cnts, clip_frames)
        if cdev.type == "cuda":
            torch.cuda.current_stream().synchronize()
        t2 = time.perf_counter()
        ids = V.replay_tracking(allrec, tracker=tracker, first_count=cnt)        # embeddings stay where the all-gather left them
        cnt += sum(1 for t in allrec if t[1].shape[0] > 0)
        seen.append(allrec)
        return t2 - t1, time.perf_counter() - t2, ids

    def one_step(step):
        """unpipelined (calibration of the components): heads -> records -> all-gather -> replay, each waited for"""
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        begin(step)
        mine, recs, cnts = finish(step)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        b, c, ids = gather_and_replay(mine, recs, cnts)
        return t1 - t0, b, c, ids

    for s_ in range(warmup):
        one_step(s_)
    # components, one step at a time (what bounds the pipelined loop below)
    calib = [one_step(warmup + s_) for s_ in range(min(steps, 6))]
    t_heads, t_coll, t_replay = [c[0] for c in calib], [c[1] for c in calib], [c[2] for c in calib]
    # a step of 8 ranks' frames replayed on this rank (the per-step costs -- one download of the boxes, one native call -- spread over
    # 8 x clip frames instead of world x clip): the calibration steps' records regrouped, a scratch tracker, 5 timed steps
    w8 = 8 * clip_frames
    flat = [r for st in seen for r in st]
    t_r8 = []
    if len(flat) >= w8:
        scratch, c8 = V.QuasiDenseEmbedTracker(**tcfg), 1
        for rep_ in range(6):
            grp = [(i, *flat[(rep_ * w8 + i) % len(flat)][1:]) for i in range(w8)]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            V.replay_tracking(grp, tracker=scratch, first_count=c8)
            if rep_:
                t_r8.append(time.perf_counter() - t0)
            c8 += sum(1 for t in grp if t[1].shape[0] > 0)
    seen.clear()
    tracker = V.QuasiDenseEmbedTracker(**tcfg)           # a new video for the timed steps (polyphonic_former_video.py:59-61)
    cnt = 1
    dist.barrier()
    torch.cuda.synchronize()
    t_all0 = time.perf_counter()
    # round 5: the step is PIPELINED -- step k + 1's heads are started (asynchronous graph replays) before step
    # k's records are all-gathered and the tracker replayed, so that a step costs max(heads, all-gather + replay), not their sum
    # round 6: the runner queues clips, so step k + 1's heads start BEFORE step k's merges / records too (the slot that frees up first;
    # heads launches run one after the other on the GPU).  It pays when a clip goes as two launches (clips of >= 4 frames: 8-frame
    # clips 794 -> 943 frames/s, 16-frame clips 838 -> 979, profiles/r06/cfg4_sweep.txt); a one-launch clip gains nothing from it
    # (2-frame clips 627 -> 524), so those keep round 5's order.  PH_CFG4_EARLY_BEGIN=0 / 1 forces either
    first = warmup + len(calib)
    _eb = os.environ.get("PH_CFG4_EARLY_BEGIN")
    early = runner is not None and (clip_frames >= 4 if _eb is None else _eb not in ("", "0"))
    begin(first)
    for s_ in range(steps):
        if early and s_ + 1 < steps:
            begin(first + s_ + 1)
        mine, recs, cnts = finish(first + s_)
        if not early and s_ + 1 < steps:
            begin(first + s_ + 1)
        _, _, ids = gather_and_replay(mine, recs, cnts)
        if collect_ids:
            ids_log.update({int(k): v.tolist() for k, v in ids.items()})
    torch.cuda.synchronize()
    dist.barrier()
    dt = D.barrier_and_max(time.perf_counter() - t_all0, cdev)
    med = lambda v: sorted(v)[len(v) // 2]
    out = {"frames_per_s": round(steps * per_step / dt, 2), "ms_per_step": round(dt / steps * 1e3, 3), "frames_per_step": per_step,
           "clip_frames_per_rank": clip_frames, "world_size": dist.get_world_size(), "backend": backend + (" (RCCL)" if backend == "nccl" else ""),
           "heads_merge_records_ms_per_step": round(D.barrier_and_max(med(t_heads), cdev) * 1e3, 3),
           "allgather_track_records_us_per_step": round(D.barrier_and_max(med(t_coll), cdev) * 1e6, 1),
           "replay_tracking_ms_per_step": round(D.barrier_and_max(med(t_replay), cdev) * 1e3, 3),
           "replay_tracking_ms_per_frame": round(D.barrier_and_max(med(t_replay), cdev) * 1e3 / per_step, 4),
           "step_pipelining": "step k+1's heads are started before step k's all-gather + tracker replay (ms_per_step ~ max of the two)",
           "precision": precision,
           "khead_onepass_timeouts": None if runner is None else runner.khead_timeouts(),
           "frame_inputs": ("borrowed: the FPN levels are read where the caller left them (no staging copy; PH_CFG4_BORROW=0 times the copying form)"
                            if runner is not None and os.environ.get("PH_CFG4_BORROW", "1") != "0" and os.environ.get("PH_VIDEO_BORROW", "1") != "0"
                            else "copied into the graph's static inputs (356 MB of traffic per frame)"),
           "frame_loop": "module API, eager launches" if runner is None else "video.VideoStreamRunner: heads replayed from one HIP graph"}
    # what the measured components project for a node of 8 ranks (the driver's 8-GPU leg, when a node is available): every rank replays
    # all 8 x clip frames of a step; the step is pipelined, so it costs max(heads of the own clip, all-gather + replay of all frames)
    h, c_, r_ = out["heads_merge_records_ms_per_step"], out["allgather_track_records_us_per_step"] * 1e-3, out["replay_tracking_ms_per_frame"]
    if t_r8:
        out["replay_tracking_ms_per_frame_at_world8_step"] = r_ = round(med(t_r8) * 1e3 / w8, 4)
    out["projected_world8_frames_per_s"] = round(w8 / max(h, c_ + r_ * w8) * 1e3, 1)
    out["projected_world8_model"] = (f"8 x {clip_frames} frames / max(heads {h} ms, all-gather {round(c_, 3)} ms + {w8} frames x replay {r_} ms) -- "
                                     "heads measured on this rank, replay on a regrouped step of 8 ranks' frames; linear scaling would be 8 x this run's frames_per_s at world 1")
    if collect_ids:
        out["track_ids"] = ids_log
    return out, pipe


def video_leg(dev, precision="bf16", frames=6):
    """BASELINE configs[2] (poly_r50_cityscapes_1x video head, 2-frame clips with tracking query match): wall time per
    1024x2048 frame of PolyphonicVideo.simple_test after extract_feat (polyphonic_former_video.py:327-405) through the
    module API -- neck + KernelHead + 3-stage decode + panoptic merge + things -> boxes -> FPN RoIAlign -> track head ->
    tracker; the shipped video head (100 + 11 queries), classification biases raised so that an un-trained network
    yields thing segments."""
    pipe, cfg, wl = _video_pipeline(dev, precision)
    H8, W8 = wl["H"] * 8, wl["W"] * 8
    g = torch.Generator().manual_seed(31)
    base = [torch.randn(1, 256, H8 // s, W8 // s, generator=g).to(dev) for s in (4, 8, 16, 32)]
    meta = [dict(img_shape=(H8, W8, 3), ori_shape=(H8, W8, 3), batch_input_shape=(H8, W8))]
    t_heads, t_assoc, t_api, nthing = [], [], [], []
    # the clip runs twice: the first pass pays every one-time cost (weight packs, kernel attributes, the graph capture, the host
    # library's first-call initialisation of the tracker's CPU ops: 90-250 ms spikes on single frames), the second one is timed.
    # `ms_per_frame` is the module API itself -- `VideoFramePipeline.simple_test`, which since round 5 replays its heads from one
    # HIP graph; `eager_launches` times the same frame through `heads` + `assoc.step` (the launches of rounds 1-4, host id map)
    for f in range(2 * frames):
        x = _video_frame(base, f, frames)
        if f == frames:
            pipe.assoc.init_tracker()          # a new clip (polyphonic_former_video.py:59-61)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = pipe.simple_test(x, meta)
        torch.cuda.synchronize()
        if f >= frames:
            t_api.append(time.perf_counter() - t0)
    assert res[0]["sem"].shape == (H8, W8)
    for f in range(2 * frames):
        x = _video_frame(base, f, frames)
        if f in (0, frames):
            pipe.assoc.init_tracker()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = pipe.heads(x, meta)[0]
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        pipe.assoc.step(x, res[2][0], res[2][1], res[4])
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if f >= frames:
            t_heads.append(t1 - t0), t_assoc.append(t2 - t1)
            nthing.append(sum(1 for s_ in res[2][1] if s_["isthing"]))
    med = lambda v: sorted(v)[len(v) // 2]
    out = {"ms_per_frame": round(med(t_api) * 1e3, 3),
           "eager_launches": {"ms_per_frame": round((med(t_heads) + med(t_assoc)) * 1e3, 3), "heads_and_merge_ms": round(med(t_heads) * 1e3, 3),
                              "association_ms": round(med(t_assoc) * 1e3, 3)},
           "thing_segments_per_frame": nthing, "frames_timed": frames,
           "precision": precision, "note": "one frame at a time (samples_per_gpu = 1 as in the reference), module API "
           "(VideoFramePipeline.simple_test: heads replayed from one HIP graph, result maps returned by the call), host wall time "
           "incl. the D2H of the sem / track / depth maps (27 MB) and the tracker"}
    # the same clip through video.VideoStreamRunner (round 4): heads from one HIP graph, the id map stays on the device, result
    # maps downloaded on a side stream under the next frame; steady-state wall time per frame, results one frame late
    try:
        from polyphonicformer_amd import video as V
        runner = V.VideoStreamRunner(pipe, meta[0])
        for rep_ in range(2):                      # first pass: capture + warm-up; second: timed
            pipe.assoc.init_tracker()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            got = 0
            for f in range(frames, 9 * frames):             # a stream of 8 x `frames` frames: steady state, fill / drain included
                r = runner.push(_video_frame(base, f, frames))
                got += r is not None
            got += len(runner.flush())
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        assert got == 8 * frames
        out["stream_runner"] = {"ms_per_frame": round(dt / (8 * frames) * 1e3, 3), "frames_timed": 8 * frames,
                                "note": "video.VideoStreamRunner: same kernels / tracker calls / results, heads + the merge's selection / "
                                        "activation / argmax replayed from one HIP graph per slot, two slots (frame t's heads run under frame "
                                        "t - 1's accept loop / association), sem / track / depth maps copied to pinned host memory on a side "
                                        "stream; results two frames late"}
    except Exception as e:
        out["stream_runner"] = {"error": repr(e)}
    # round 6: the same stream with k frames per heads launch (batch-invariant heads: the same result maps), two launches in flight
    for k in (4, 8):
        try:
            from polyphonicformer_amd import video as V
            runner = V.VideoStreamRunner(pipe, meta[0], frames_per_launch=k)
            if runner._launch_size() != k:
                out[f"stream_runner_{k}_frames_per_launch"] = {"skipped": "the heads' grade is not batch invariant (two-pass KernelHead)"}
                continue
            for rep_ in range(2):
                pipe.assoc.init_tracker()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                got = 0
                nf = 24 * frames                 # (fill + drain are one launch each: 24 x 6 frames = 36 / 18 launches)
                for f in range(frames, frames + nf):
                    r = runner.push(_video_frame(base, f, frames))
                    got += r is not None
                got += len(runner.flush())
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            assert got == nf
            out[f"stream_runner_{k}_frames_per_launch"] = {"ms_per_frame": round(dt / nf * 1e3, 3), "frames_timed": nf,
                                                          "note": f"VideoStreamRunner(frames_per_launch={k}): results up to {2 * k} frames late, bit-identical maps"}
            del runner
        except Exception as e:
            out[f"stream_runner_{k}_frames_per_launch"] = {"error": repr(e)}
    return out


def train_leg(wl, dev, world, B=2, steps=4):
    """secondary number: one TRAINING step of the path per image batch -- both heads forward in training mode
    from the three post-neck maps, Hungarian assignment (host, scipy), targets, losses, backward to every parameter and to the
    maps (polyphonicformer_amd/train.py), gradient all-reduce over the process group when there is more than one rank."""
    from polyphonicformer_amd.registry import HEADS
    from polyphonicformer_amd import train as T
    import polyphonicformer_amd.kernel_head, polyphonicformer_amd.kernel_update  # noqa: F401,E401
    import polyphonicformer_amd.kernel_update_head, polyphonicformer_amd.kernel_updator  # noqa: F401,E401
    L, nt, ns, H, W = wl["n_thing"] + wl["n_stuff"], wl["n_thing"], wl["n_stuff"], wl["H"], wl["W"]
    cost = dict(cls_cost=dict(type='FocalLossCost', weight=2.0), dice_cost=dict(type='DiceCost', weight=4.0, pred_act=True),
                mask_cost=dict(type='MaskCost', weight=1.0, pred_act=True))
    tc = lambda extra: dict(assigner=dict(type='MaskHungarianAssignerWithDepth', **cost, **extra), sampler=dict(type='MaskPseudoSampler'),
                            pos_weight=1.)
    torch.manual_seed(2)
    rpn = HEADS.build(dict(type="KernelHead", num_proposals=wl["Nq"], num_classes=L, num_thing_classes=nt, num_stuff_classes=ns,
                           cat_stuff_mask=True, feat_downsample_stride=2, feat_refine=False, use_binary=True, proposal_feats_with_obj=True,
                           loss_rank=dict(type="CrossEntropyLoss", use_sigmoid=False, loss_weight=0.1),
                           loss_seg=dict(type="FocalLoss", use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0),
                           loss_mask=dict(type="CrossEntropyLoss", use_sigmoid=True, loss_weight=1.0),
                           loss_dice=dict(type="DiceLoss", loss_weight=4.0),
                           loss_depth=dict(type="DepthLoss", loss_weight=5.0, depth_act_mode="sigmoid"), train_cfg=tc({})))
    scfg = stage_cfg(L, nt, ns, wl["F"])
    scfg.update(loss_rank=dict(type="CrossEntropyLoss", use_sigmoid=False, loss_weight=0.1),
                loss_cls=dict(type="FocalLoss", use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=2.0),
                loss_dice=dict(type="DiceLoss", loss_weight=4.0), loss_depth=dict(type="DepthLoss", loss_weight=5.0, depth_act_mode="sigmoid"))
    depth_cost = dict(depth_cost=dict(type='DepthCost', weight=0., loss_fn=dict(type='DepthMatchLoss', loss_weight=1.), depth_act_mode='sigmoid'))
    roi = HEADS.build(dict(type="KernelUpdateIterHead", num_stages=wl["S"], assign_stages=wl["S"], stage_loss_weights=[1] * wl["S"],
                           num_proposals=wl["Nq"], num_thing_classes=nt, num_stuff_classes=ns, mask_head=scfg, train_cfg=tc(depth_cost)))
    rpn.init_weights()
    roi.init_weights()
    rpn.to(dev)
    roi.to(dev)
    step = T.TrainStep(rpn, roi)
    g = torch.Generator().manual_seed(11)
    feats = [torch.randn(B, 256, H, W, generator=g).relu().to(dev) for _ in range(3)]
    H2, W2 = 2 * H, 2 * W
    gts = []
    for b in range(B):                       # ~20 instances and half of the stuff classes per image, at the assign stride
        G = 20
        cy, cx = torch.rand(G, generator=g) * H2, torch.rand(G, generator=g) * W2
        r = 8 + torch.rand(G, generator=g) * 40
        yy, xx = torch.arange(H2)[None, :, None], torch.arange(W2)[None, None, :]
        masks = (((yy - cy[:, None, None]) ** 2 + (xx - cx[:, None, None]) ** 2) < r[:, None, None] ** 2).float()
        present = torch.randperm(ns, generator=g)[: ns // 2].sort()[0]
        sem = (torch.rand(len(present), H2 // 16, W2 // 16, generator=g) > 0.6).float()
        sem = torch.nn.functional.interpolate(sem[None], size=(H2, W2), mode="nearest")[0]
        depth = torch.rand(H2, W2, generator=g) * 79.0 + 0.5
        gts.append(dict(masks=masks.to(dev), labels=torch.randint(0, nt, (G,), generator=g).to(dev), sem_seg=sem.to(dev),
                        sem_cls=(present + nt).to(dev), depth=depth.to(dev)))
    metas = [dict(img_shape=(H * 8, W * 8, 3), ori_shape=(H * 8, W * 8, 3), batch_input_shape=(H * 8, W * 8))] * B
    gd = torch.stack([x["depth"][None] for x in gts])
    args = (feats, metas, [x["masks"] for x in gts], [x["labels"] for x in gts], [x["sem_seg"] for x in gts], [x["sem_cls"] for x in gts], gd)

    def one(backward=True):
        for p in step.parameters():
            p.grad = None
        return step.forward_backward(*args, backward=backward)

    one()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        losses, total, _ = one()
    torch.cuda.synchronize(dev)
    full = (time.perf_counter() - t0) / steps
    t0 = time.perf_counter()
    for _ in range(steps):
        one(backward=False)
    torch.cuda.synchronize(dev)
    fwd = (time.perf_counter() - t0) / steps
    nparam = sum(p.numel() for p in step.parameters())
    return {"images_per_step_per_gpu": B, "ms_per_step": round(full * 1e3, 2), "forward_only_ms": round(fwd * 1e3, 2),
            "images_per_s": round(world * B / full, 2), "objective": round(float(total), 3), "parameters": nparam,
            "grad_allreduce": f"{len(step.buckets.buckets)} bucket(s) over {world} rank(s)" + ("" if world > 1 else " (nothing sent)"),
            "workload": f"{H * 8}x{W * 8}, stride-8 maps {H}x{W}, losses at stride 4, N={wl['Nq']}+{ns}, S={wl['S']}, 20 instances / image",
            "note": "forward (training mode) + Hungarian assignment on the host + targets + losses + backward to all parameters and "
                    "the three post-neck maps; fp32, wall time incl. host work"}


def panoptic_leg(wl, head, plan, dev):
    """get_panoptic (a7, SURVEY 8d: reported separately) on ONE frame of the step's outputs; host wall time,
    including the D2H of the int32 id map and the two fp32 depth maps that the reference API returns as numpy"""
    from polyphonicformer_amd import panoptic as Pn
    from polyphonicformer_amd.registry import ConfigDict
    # An un-trained head accepts nothing (fc_cls.bias = -4.6 -> every score < instance_score_thr; its random masks overlap
    # heavily, so the 0.6 overlap test fails too) and the leg would time a frame whose accept loop and depth paste are empty
    # (VERDICT r03 weak 10).  The masks and depth maps are the step's real outputs; the class scores are synthetic -- 40 thing
    # queries and every stuff class above the threshold -- and overlap_thr is 0 (a segment is accepted when it wins a pixel),
    # as the video leg arranges it: tens of accepted segments, the workload of a trained model's frame.
    head.test_cfg = ConfigDict(max_per_img=wl["Nq"], merge_stuff_thing=dict(overlap_thr=0.0, instance_score_thr=0.3))
    o = plan.outputs()
    H, W = wl["H"], wl["W"]
    N, L, nt = wl["Nq"] + wl["n_stuff"], wl["n_thing"] + wl["n_stuff"], wl["n_thing"]
    g = torch.Generator().manual_seed(17)
    cls = torch.full((N, L), 0.02)
    hot = torch.randperm(wl["Nq"], generator=g)[:40]
    cls[hot, torch.randint(0, nt, (40,), generator=g)] = 0.35 + 0.6 * torch.rand(40, generator=g)
    sidx = torch.arange(wl["n_stuff"])
    cls[wl["Nq"] + sidx, nt + sidx] = 0.4 + 0.5 * torch.rand(wl["n_stuff"], generator=g)
    cls = cls.to(dev)
    meta = dict(img_shape=(H * 8, W * 8, 3), ori_shape=(H * 8, W * 8, 3), batch_input_shape=(H * 8, W * 8))
    d0 = torch.randn(1, 2 * H, 2 * W, device=dev)
    ts = []
    for _ in range(4):
        torch.cuda.synchronize()
        t = time.perf_counter()
        r = Pn.get_panoptic(head, cls, o["mask_up"][0], o["depth_up"][0], d0, meta)
        ts.append((time.perf_counter() - t) * 1e3)
    return {"ms_per_frame": round(min(ts[1:]), 3), "segments": len(r[2][1]), "things": sum(1 for s_ in r[2][1] if s_["isthing"]),
            "output": f"int32 {H * 8}x{W * 8} id map + 2 fp32 depth maps on the host",
            "inputs": "the step's mask / depth logits of frame 0; synthetic class scores (40 thing queries + all stuff classes above "
                      "instance_score_thr), overlap_thr 0 so that an un-trained network's overlapping masks are accepted"}


def cpu_baseline(wl, head, budget_s=16.0, all_cores=True, workload_note="the same workload (1024x2048, N=153, S=3)"):
    """the oracle (CPU restatement of the reference path) on this box's host cores, bounded sample"""
    from oracle import poly_oracle as O
    sd = {k: v.detach().cpu() for k, v in head.state_dict().items()}
    inp = synth_inputs(wl, 1, 1)
    # SURVEY 8d asks for k = 1 and k = all cores; 16 threads is the best of a {8,16,32,64,128}-thread sweep on the GPU
    # box's 2 x EPYC 9575F (tools/cpu_sweep.py; more threads are slower: the path is memory/latency bound on CPU) and is
    # the `value` reported.  The sample budget is split 2 : 1 : 1 over the three settings.
    S = wl["S"]
    allc = os.cpu_count() or 1

    def timed(k, budget):
        torch.set_num_threads(k)
        with torch.no_grad():
            t0 = time.time()
            O.iter_head_mask_preds(sd, S, inp["x"], inp["k0"], inp["m0"], inp["q0"], inp["dfe"])   # warm-up
            warm = time.time() - t0
            n, t1 = 0, time.time()
            while n < 200 and (time.time() - t1) < budget:
                O.iter_head_mask_preds(sd, S, inp["x"], inp["k0"], inp["m0"], inp["q0"], inp["dfe"])
                n += 1
            dt = (time.time() - t1) / max(n, 1)
        return (n, dt) if n else (1, warm)

    ncores = min(16, allc)
    n, dt = timed(ncores, budget_s * 0.5)
    n1, dt1 = timed(1, budget_s * 0.25)
    # BASELINE.md section 4 asks for the all-cores figure in the default line.  All 256 hardware threads of the GPU box oversubscribe
    # these small ops (~45 s per frame: --all-legs runs it once); the default line carries min(64, all) threads -- as many
    # as there are physical cores on one socket's worth of the box -- on a 3-second sample
    nmany = min(64, allc)
    many = None
    if nmany != ncores:
        nm, dtm = timed(nmany, 3.0)
        many = {"value": 1.0 / dtm, "cores": nmany, "frames": nm}
    if allc != ncores and all_cores:      # hundreds of threads on these small ops are pathologically slow (~45 s per frame on 256): ONE frame
        torch.set_num_threads(allc)
        with torch.no_grad():
            t0 = time.time()
            O.iter_head_mask_preds(sd, S, inp["x"], inp["k0"], inp["m0"], inp["q0"], inp["dfe"])
        na, dta = 1, time.time() - t0
    else:
        na, dta = (n, dt) if allc == ncores else (0, float("inf"))
    torch.set_num_threads(ncores)
    extra = {}
    try:        # the assigner's cost matrices (SURVEY 8f N4 first part) by the oracle, same sizes as the `hungarian_assign` leg
        from oracle import assign_oracle as AO
        g = torch.Generator().manual_seed(5)
        zc = torch.randn(100, 128, 256, generator=g) * 2
        tc = (torch.rand(40, 128, 256, generator=g) > 0.7).float()
        vc = (torch.rand(128, 256, generator=g) > 0.1).float()
        t0 = time.perf_counter()
        for _ in range(3):
            AO.dice_cost(zc, tc, vc)
            AO.mask_cost(zc, tc, vc)
        extra["assign_costs_ms_per_image"] = round((time.perf_counter() - t0) / 3 * 1e3, 2)
    except Exception as e:
        extra["assign_costs_ms_per_image"] = repr(e)
    out = dict(value=1.0 / dt, unit="frames/s", cores=ncores, kind="port", **extra,
               one_thread={"value": 1.0 / dt1, "cores": 1, "frames": n1},
               sample=f"{n} frame(s) of {workload_note}, fp32, B=1, after 1 warm-up; + {n1} frame(s) on 1 thread")
    if many:
        out["many_cores"] = many
        out["sample"] += f"; + {many['frames']} frame(s) on {nmany} threads"
    if na:
        out["all_cores"] = {"value": 1.0 / dta, "cores": allc, "frames": na}
        out["sample"] += f" and {na} on all {allc} hardware threads"
    else:
        out["all_cores"] = f"not run by default ({allc} hardware threads take ~45 s per frame: --all-legs; profiles/r03/bench_all_legs.json has 0.022 frames/s) -- `many_cores` is the default line's wide figure"
    return out


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(ngpus):
    """`python bench.py --gpus N` without an outer launcher: this process becomes rank 0 and starts ranks 1..N-1 as
    copies of itself, one process per GPU, rendezvous on 127.0.0.1 (what the reference's tools/dist_test.sh /
    tools/dist_train.sh:11-32 do with torch.distributed.launch).  Returns the worker processes."""
    import subprocess
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if os.environ.get("PH_DIST_BACKEND", "nccl") == "nccl" and ngpus > ndev:
        raise SystemExit(f"--gpus {ngpus} but only {ndev} GPU(s) are visible on this node")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE=str(ngpus),
               PH_BENCH_SELF_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    workers = []
    for r in range(1, ngpus):
        workers.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                        env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.DEVNULL))
    os.environ.update(env, RANK="0", LOCAL_RANK="0")
    return workers


def track_allgather_leg(dev, world, backend, frames_per_rank=2, iters=30):
    """cfg4's one exchange: every rank contributes `frames_per_rank` padded track-record blocks
    [frames, 100, 262] fp32 (+ an int64 [frames, 2] (frame id, count) vector); ONE all-gather pair per step hands every
    rank all records.  Timed on the process group the bench runs on ("nccl" = RCCL over xGMI; world size asserted)."""
    import torch.distributed as dist
    from polyphonicformer_amd import dist as D
    assert dist.is_initialized() and dist.get_world_size() == world, "process group does not span --gpus ranks"
    rank = dist.get_rank()
    cdev = dev if backend == "nccl" else torch.device("cpu")
    g = torch.Generator().manual_seed(77 + rank)
    fids = [rank * frames_per_rank + i for i in range(frames_per_rank)]
    recs, cnts = [], []
    for _ in fids:
        n = int(torch.randint(20, 100, (1,), generator=g))
        r, n = D.pack_track_records(torch.rand(n, 5, generator=g).to(cdev), torch.randint(0, 8, (n,), generator=g).to(cdev),
                                    torch.randn(n, 256, generator=g).to(cdev))
        recs.append(r)
        cnts.append(n)
    blk = torch.stack(recs, 0)
    meta = torch.tensor([[f, n] for f, n in zip(fids, cnts)], dtype=torch.int64, device=cdev)
    all_blk = torch.empty((world * frames_per_rank,) + tuple(blk.shape[1:]), dtype=blk.dtype, device=cdev)
    all_meta = torch.empty((world * frames_per_rank, 2), dtype=torch.int64, device=cdev)

    def sync():
        if cdev.type == "cuda":
            torch.cuda.synchronize()

    def collective():
        dist.all_gather_into_tensor(all_blk, blk)
        dist.all_gather_into_tensor(all_meta, meta)

    for _ in range(5):
        collective()
    sync()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(iters):
        collective()
    sync()
    t_coll = (time.perf_counter() - t0) / iters
    out = D.allgather_track_records(fids, recs, cnts, frames_per_rank)          # the full call incl. the host unpack
    sync()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(iters):
        out = D.allgather_track_records(fids, recs, cnts, frames_per_rank)
    sync()
    t_call = (time.perf_counter() - t0) / iters
    ok = [t[0] for t in out] == list(range(world * frames_per_rank)) and torch.equal(out[fids[0]][3], recs[0][:cnts[0], 6:])
    t_coll = D.barrier_and_max(t_coll, cdev)
    t_call = D.barrier_and_max(t_call, cdev)
    return {"backend": backend + (" (RCCL)" if backend == "nccl" else ""), "world_size": dist.get_world_size(),
            "frames_per_rank": frames_per_rank, "payload_bytes_per_rank": int(blk.numel() * 4 + meta.numel() * 8),
            "collective_us_per_step": round(t_coll * 1e6, 1), "allgather_track_records_us_per_call": round(t_call * 1e6, 1),
            "payload_round_trip_exact": bool(ok),
            "note": "all_gather_into_tensor of [frames,100,262] fp32 records + [frames,2] int64 meta per rank, max over ranks; "
                    "the second number adds the host-side unpack (one D2H of the meta vector) of dist.allgather_track_records"}


def cfg4_main(args, dev, world, rank, backend, json_fd):
    """`python bench.py --workload cfg4 --gpus N`: BASELINE configs[3] end to end (see cfg4_run); rank 0 prints the JSON line
    with frames/s (all ranks' frames / the max-over-ranks time), the collective's time, roofline and cpu_baseline."""
    import torch.distributed as dist
    prec = args.precision if args.precision in ("bf16", "fp32") else "fp16"       # the contract grade of neck / heads
    run, pipe = cfg4_run(dev, world, rank, backend, precision=prec, clip_frames=args.clip_frames, steps=args.steps, warmup=args.warmup)
    if rank != 0:
        return
    wl = WORKLOADS["cfg3"]
    N = wl["Nq"] + wl["n_stuff"]
    plan = next(iter(pipe.roi_head._plans.values()))                 # the decode plan the frames ran on: one frame per launch
    times, counts = kernel_breakdown(plan)
    ab = algorithmic_bytes(plan, "pool")
    achieved = ab / (times["pool"] * 1e-3) / 1e9
    res = {"metric": f"frames/sec video head (heads + merge + association), {wl['H'] * 8}x{wl['W'] * 8} N={N} S={wl['S']}, clips sharded 1/GPU",
           "value": run["frames_per_s"], "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": run["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": {"fp16": "fp16 (planes, dynamic kernels, logits; hi/lo bf16 query GEMMs)", "bf16": "bf16", "fp32": "bf16x3 (fp32-grade split)"}[prec],
           "data": "synthetic", "host_threads": torch.get_num_threads(),
           "config": {"workload": f"cfg4: poly_r50 video head, {wl['H'] * 8}x{wl['W'] * 8}, N={N}, S={wl['S']}, {args.clip_frames}-frame clip per "
                                  f"GPU and step, one frame per launch (samples_per_gpu = 1 as in the reference), module API, random-init weights",
                      "frames_per_step_per_gpu": args.clip_frames,
                      "parallelism": f"clips sharded over {world} GPU(s); ONE all-gather of the track records per step on "
                                     f"{run['backend']}, tracker replayed in frame order on every rank"},
           "cfg4": run,
           "roofline": {"bound": "hbm", "kernel": "pool", "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
                        "frac": round(achieved / 8000.0, 4), "traffic": None, "algorithmic_bytes_per_launch": ab,
                        "avg_launch_ms": round(times["pool"], 4), "frames_per_launch": plan.B,
                        "note": "the path's HBM-dominant contraction at THIS workload's launch geometry: one frame per launch, i.e. "
                                "latency bound (the reference's video loop is one frame at a time); cfg2's batched launches are the "
                                "default workload"},
           "kernels_ms": {k: round(v, 4) for k, v in times.items()}}
    if not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(wl, pipe.roi_head, all_cores=False, workload_note=f"the decode part, {wl['H'] * 8}x{wl['W'] * 8}, N={N}, S={wl['S']}")
    os.write(json_fd, (json.dumps(res) + "\n").encode())


HOST_THREADS = 16


def host_thread_policy():
    """Host thread policy of THIS process (the bench, not the library): torch's intra-op pool defaults to one thread per
    hardware thread -- 128 here on a 256-thread host -- whose workers keep spinning after every parallel region.  Inside a
    container with a CPU quota that burns the quota and the kernel throttles the whole process for the rest of the 100 ms
    period: 80-95 ms stalls in ANY host call (seen as spikes in `video_cfg3` / cfg4 frames: aten::tril of an 8 x 8 tensor
    taking 90 ms; none with one thread, round 3).  16 threads is also the best setting of the CPU baseline sweep.  Returns the
    setting for the JSON line."""
    n = min(HOST_THREADS, torch.get_num_threads())
    torch.set_num_threads(n)
    return n


def main():
    host_threads = host_thread_policy()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--frames", type=int, default=128, help="frames per step per GPU (round 6: 128 = 4 parts of 32 -- the fused conv + pooling launch then has exactly one workgroup per CU; 96 until then)")
    ap.add_argument("--precision", default="mixed16", choices=["bf16", "mixed", "mixed16", "fp16", "fp32"],
                    help="engine.MODES: bf16 = one bf16 plane everywhere (fast, ~6e-3 per stage); mixed = bf16 feature planes as "
                         "given, fp32-grade arithmetic on them (1.2e-5 per stage on identical inputs), fp16 logits out; mixed16 = "
                         "the same with ONE fp16 plane of dynamic kernels and of the attention / FFN / tower half of the query side (6e-4 per stage: the cheapest mode inside the 1e-3 "
                         "contract on bf16 inputs, the default); fp16 = fp16 planes / "
                         "kernels / logits (cfg5), fp32-grade query side; fp32 = every operand hi + lo (parity grade)")
    ap.add_argument("--workload", default="cfg2", choices=list(WORKLOADS) + ["cfg4"],
                    help="cfg2 (default, BASELINE's metric) / cfg3 / cfg5 / tiny: simple_test_mask_preds at that geometry; cfg4: the "
                         "video head with clips sharded one per GPU and the track records all-gathered (configs[3])")
    ap.add_argument("--input-dtype", default="auto", choices=["auto", "fp32", "bf16", "fp16"],
                    help="dtype of the x_feats / depth_feats inputs resident in HBM; auto = the precision's own "
                         "(bf16 NCHW tensors are the kernels' plane format, fp32 ones go through the ingest kernel)")
    ap.add_argument("--mask-logits", default="auto", choices=["auto", "fp32"],
                    help="dtype of the initial mask logits resident in HBM: auto = the mode's 16-bit logit format when the features are "
                         "16-bit (what a KernelHead at that grade hands over; SURVEY 8d counts them at e_f = 2), fp32 = round 4's tensor")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--streams", type=int, default=4, choices=[1, 2, 3, 4, 6, 8],
                    help="n > 1: n part-batches on n skewed HIP streams, one HIP graph (engine.DualDecodePlan)")
    ap.add_argument("--all-legs", action="store_true",
                    help="also run the secondary legs (all five precision modes, cfg5, the parity / fast pairs of a1 + a6, the neck, "
                         "the whole head from the FPN levels, video cfg3, the assigner, the training step, the 256-thread CPU "
                         "point): ~10 minutes; the default run takes about two")
    ap.add_argument("--clip-frames", type=int, default=2, help="cfg4: frames of a rank's clip per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-head", action="store_true")
    ap.add_argument("--no-neck", action="store_true")
    ap.add_argument("--mask-bias", type=float, default=0.0, help="added to the initial mask logits (-2: sparse masks)")
    args = ap.parse_args()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # stdout carries exactly ONE line, the JSON: everything else that writes to fd 1 from here on (RCCL prints a version
    # banner through C stdio when its communicator comes up) is sent to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    workers = []
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        workers = self_launch(args.gpus)        # plain `python bench.py --gpus N`: start the other N-1 ranks ourselves
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with as many ranks as GPUs")
    ndev = torch.cuda.device_count()
    backend = os.environ.get("PH_DIST_BACKEND", "nccl")        # "gloo": lets 2 ranks share ONE GPU (path test only)
    if world > 1 and backend == "nccl" and local_rank >= ndev:
        raise SystemExit(f"LOCAL_RANK {local_rank} but only {ndev} GPU(s) visible")
    dev = torch.device("cuda", local_rank % ndev)
    torch.cuda.set_device(dev)
    # a process group always exists (world 1 included), so that the one collective of the path is timed on RCCL at every N
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world == 1:
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)

    if args.workload == "cfg4":
        cfg4_main(args, dev, world, rank, backend, json_fd)
        dist.barrier()
        dist.destroy_process_group()
        rc = 0
        for w in workers:
            rc |= w.wait()
        if rc:
            raise SystemExit(f"a self-launched rank exited with status {rc}")
        return
    wl = WORKLOADS[args.workload]
    B = args.frames
    out_dtype = {"bf16": torch.bfloat16, "mixed": torch.float16, "mixed16": torch.float16, "fp16": torch.float16, "fp32": torch.float32}[args.precision]
    head = build_head(wl, args.precision, out_dtype, dev)
    N = wl["Nq"] + wl["n_stuff"]
    plan = head._plan(B, N, wl["H"], wl["W"], dev)      # single-stream plan (also used for the per-kernel timings)
    inp = synth_inputs(wl, B, seed=1234 + rank, mask_bias=args.mask_bias)         # each rank: its own frames
    in_dt = args.input_dtype if args.input_dtype != "auto" else {"bf16": "bf16", "mixed": "bf16", "mixed16": "bf16", "fp16": "fp16", "fp32": "fp32"}[args.precision]
    gin = [inp[k].to(dev) for k in ("x", "dfe", "k0", "q0", "m0")]
    if in_dt in ("bf16", "fp16"):
        tdt = torch.bfloat16 if in_dt == "bf16" else torch.float16
        gin[0], gin[1] = gin[0].to(tdt), gin[1].to(tdt)
        # round 5: the initial mask logits arrive 16-bit too -- cfg2 is the bf16 configuration, SURVEY 8d's algorithmic bytes count
        # them as N * HW * e_f with e_f = 2, and they are what KernelHead hands over at this grade (its `logit_dtype`); in the
        # logits' own 16-bit format (the mode's output dtype).  --mask-logits fp32 restores round 4's fp32 tensor
        if args.mask_logits == "auto" and out_dtype != torch.float32:
            gin[4] = gin[4].to(out_dtype)
    plan.set_inputs(*gin)
    runner = plan
    if args.streams > 1:
        from polyphonicformer_amd.engine import DualDecodePlan
        runner = DualDecodePlan(plan.packs, B, N, wl["H"], wl["W"], plan.mode, out_dtype, dev, parts=args.streams)
        runner.set_inputs(*gin)
    if args.no_graph:
        step = runner.run
    else:
        runner.capture()
        step = runner.replay

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # set-up, outside the W + K protocol: a freshly allocated box clocks up and faults its pages in during the first
    # replays (measured: 12.5 k frames/s with --warmup 2 --steps 5 alone, 13.3 k after these); every replay recomputes all
    # (round 6: at least a second of them -- the first process on a fresh box measured 14.5 k where its later runs measured 15.5 k)
    t_setup = time.perf_counter()
    n_setup = 0
    while n_setup < 12 or time.perf_counter() - t_setup < 1.0:
        step()
        n_setup += 1
        if n_setup % 16 == 0:
            torch.cuda.synchronize()
    barrier()
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    from polyphonicformer_amd.dist import barrier_and_max
    dt = barrier_and_max(dt, dev if (world == 1 or backend == "nccl") else torch.device("cpu"))   # MAX over ranks
    fps = world * B * args.steps / dt

    # the one exchange of the path (video track records), outside the timed region, every rank takes part
    try:
        tag = {fpr: track_allgather_leg(dev, world, backend, frames_per_rank=fpr) for fpr in (2, 8)}
    except Exception as e:
        tag = {"error": repr(e)}

    if rank == 0:
        # per-launch timings in the geometry the timed region launches: one half-batch plan when two streams are used
        kplan = runner.halves[0] if args.streams > 1 else plan
        nplans = args.streams if args.streams > 1 else 1
        times, counts = kernel_breakdown(kplan)
        per_step = {k: times[k] * counts.get(k, 0) * nplans for k in times}
        roof = roofline_of(kplan, times, counts, nplans)
        dom = roof["kernel"]
        # HBM bytes per launch come from separate rocprofv3 --pmc passes (FETCH_SIZE x2 + WRITE_SIZE, tools/pmc_summary.py),
        # which cannot run inside this process: the committed summary is quoted, labelled as such, and only when the launch
        # geometry is the profiled one
        traffic, traffic_src = None, None
        members = ["dynconv_up2_mask", "dynconv_up2_depth"] if dom == "dynconv_up2" else [dom]
        for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
            try:
                with open(os.path.join(REPO, "profiles", rnd, "pmc_traffic.json")) as f:
                    pt = json.load(f)
                if pt["frames_per_launch"] == kplan.B and args.workload == "cfg2" \
                        and args.precision == pt.get("mode", "bf16") and all(m in pt["kernels"] for m in members):
                    traffic = sum(pt["kernels"][m]["hbm_bytes_per_launch"] for m in members) // len(members)
                    traffic_src = f"profiles/{rnd}/pmc_traffic.json (rocprofv3 --pmc passes of this command, not measured in this run; per launch, " \
                                  f"mean of {' + '.join(members)})"
                    break
            except Exception:
                continue
        # what the GRAPH replays: per-kernel durations inside the multi-stream step and the overlap efficiency come
        # from a rocprofv3 kernel trace of this command (tools/timeline.py), which cannot run inside this process either: quoted
        # from the committed summary of this round, labelled as such
        in_graph = None
        try:
            with open(os.path.join(REPO, "profiles", "r06", "timeline_4streams.json")) as f:
                tl = json.load(f)
            if tl["parts"] == nplans and args.workload == "cfg2" and args.precision == "mixed16" and B == tl.get("frames_per_step", 96):
                in_graph = {"source": "profiles/r06/timeline_4streams.json (tools/timeline.py on a rocprofv3 --kernel-trace of this command; not "
                                      "measured in this run)",
                            "wall_us_per_step": tl["wall_us_per_step"], "overlap_efficiency": tl.get("overlap_efficiency"),
                            "query_time_not_hidden_us_per_part": tl.get("query_time_not_hidden_us_per_part"),
                            "query_only_us_per_step": tl["query_exposed_us"], "idle_us_per_step": tl["idle_us"],
                            "isolated_kernel_us": tl.get("isolated", {}).get("median_us"),
                            "in_step_mean_kernel_us": {k: v["mean_us"] for k, v in tl["in_step"].items()}}
        except Exception:
            in_graph = None
        res = {
            "metric": "frames/sec kernel-update+mask fwd, 1024x2048 N=153 S=3" if args.workload == "cfg2" else
                      f"frames/sec kernel-update+mask fwd, {wl['H'] * 8}x{wl['W'] * 8} N={N} S={wl['S']}", "value": round(fps, 2),
            "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"bf16": "bf16", "mixed": "bf16 (feature planes as given; hi/lo bf16 query GEMMs and dynamic kernels, fp16 logits)",
                      "mixed16": "bf16 (feature planes as given; query side: updator half hi/lo bf16, attention / FFN / towers one fp16 plane; one fp16 plane of dynamic kernels, fp16 logits)",
                      "fp16": "fp16 (planes, dynamic kernels, logits; query side: updator half hi/lo bf16, the rest one fp16 plane)",
                      "fp32": "bf16x3 (fp32-grade split)"}[args.precision],
            "data": "synthetic", "host_threads": host_threads,
            "config": {"workload": f"{args.workload}: KernelUpdateIterHead.simple_test_mask_preds, "
                                   f"{wl['H'] * 8}x{wl['W'] * 8}, stride-8 {wl['H']}x{wl['W']}, N={N}, S={wl['S']}, "
                                   f"L={wl['n_thing'] + wl['n_stuff']}, random-init weights",
                       "frames_per_step_per_gpu": B, "hip_graph": not args.no_graph, "streams": args.streams,
                       "feature_input_dtype": in_dt, "mask_logit_input_dtype": str(gin[4].dtype).replace("torch.", ""), "output_dtype": str(out_dtype),
                       "parity_inputs": {"bf16": "bf16-rounded features on both sides (oracle fed the bf16 tensors the device reads); 6.6e-3 per stage: outside the 1e-3 contract",
                                         "mixed": "bf16-rounded features on both sides (oracle fed the bf16 tensors the device reads): 1.2e-5 per stage",
                                         "mixed16": "bf16-rounded features on both sides (oracle fed the bf16 tensors the device reads): 5.3e-4-6.6e-4 per stage; "
                                                    "against UNROUNDED fp32 features this mode is gated at 3e-3 -- the `fp16` mode (precision_modes) meets 1e-3 there",
                                         "fp16": "unrounded fp32 features into the oracle, fp16-rounded into the device: <= 6.6e-4 per stage",
                                         "fp32": "fp32 features on both sides: 1.3e-5 per stage"}[args.precision],
                       "parallelism": f"frames sharded over {world} GPU(s), no data-path collective"},
            "roofline": dict(roof, traffic=traffic, traffic_source=traffic_src,
                             achievable_GBps_from_profiles={"pure_read": 5800.0, "pure_fill": 6900.0, "copy": 5600.0},
                             achievable_source="profiles/r01/dmabw_ring_microbenchmark.txt (compute-free LDS-DMA ring read), "
                                               "profiles/r05/writebw_yardstick.txt (fill / copy); not measured in this run"),
            # SURVEY 8d "Reporting": whole-path algorithmic rates of the timed step (B_alg / F_alg per frame incl. the
            # per-stage weight stream amortised over the frames of a launch)
            "algorithmic": algorithmic_rates(wl, N, kplan.B, fps / world, args.precision),
            "per_stage_ms": {"frames": kplan.B, "non_final": round(times["pool"] + times["query_pre"] + times["query_post"] +
                                                                   times.get("dynconv_bits", times.get("dynconv_poolx", 0.0)), 4),
                             "final_incl_upsample": round(times["pool"] + times["query_pre"] + times["query_post"] +
                                                          (times["dynconv_up2_mask"] + times["dynconv_up2_depth"] if "dynconv_up2_mask" in times
                                                           else 2 * times["dynconv_logits"] + 2 * times["upsample2x"]), 4)},
            "kernels_ms": {k: round(v, 4) for k, v in times.items()},
            "kernels_ms_per_step": {k: round(v, 4) for k, v in per_step.items()},
            "kernels_ms_note": f"HIP-event timings of one part's launches issued eagerly on one stream (an event after every launch: 12-20 % above the "
                               f"kernel-trace durations); the step itself replays {nplans} part(s) on {nplans} skewed streams from one graph, whose kernels "
                               f"overlap -- `in_graph` holds what the graph replays",
            "in_graph": in_graph,
            "track_allgather": tag,
        }
        full = args.all_legs
        if world == 1 and full and in_dt in ("bf16", "fp16") and not args.no_kernel_head:
            # the same step when the features arrive as fp32 NCHW (the reference's dtype) and go through the ingest kernel
            try:
                gin32 = [inp[k].to(dev) for k in ("x", "dfe", "k0", "q0", "m0")]
                runner.set_inputs(*gin32)
                if not args.no_graph:
                    runner.capture()
                t32 = time_op(step, 10)
                res["fp32_feature_inputs"] = {"value": round(B / (t32 * 1e-3), 2), "unit": "frames/s", "ms_per_step": round(t32, 4),
                                              "note": "same step + 2 ingest launches (fp32 NCHW -> bf16 planes) inside the timed region"}
            except Exception as e:
                res["fp32_feature_inputs"] = {"error": repr(e)}
        if world == 1 and not args.no_kernel_head and args.workload == "cfg2":
            # the same function in the other precision modes (engine.MODES), each with inputs resident in its own plane
            # format and its own output dtype; per-stage error against the fp32 oracle from tests/test_gpu_configs.py
            err_note = {"bf16": "6.6e-3 per stage (identical bf16 inputs): the fast mode, outside the 1e-3 contract",
                        "mixed": "1.2e-5 per stage on identical bf16 inputs (1e-3 contract met)",
                        "mixed16": "5.3e-4-6.6e-4 per stage on identical bf16 inputs (1e-3 contract met)",
                        "fp16": "<= 6.6e-4 per stage against unrounded fp32 inputs (1e-3 contract met)",
                        "fp32": "1.3e-5 per stage against fp32 inputs (parity grade)"}
            res["precision_modes"] = {args.precision: {"value": round(fps, 2), "unit": "frames/s", "per_stage_rel_err": err_note[args.precision]}}
            for mode in (("bf16", "mixed", "mixed16", "fp16", "fp32") if full else ("fp16", "mixed16")):
                if mode == args.precision:
                    continue
                try:
                    res["precision_modes"][mode] = dict(mode_leg(wl, mode, dev, B if mode != "fp32" else 32, args.streams if mode != "fp32" else 2),
                                                        per_stage_rel_err=err_note[mode])
                except Exception as e:
                    res["precision_modes"][mode] = {"error": repr(e)}
        if world == 1 and not args.no_kernel_head and args.workload == "cfg2":
            try:        # BASELINE configs[4] as specified: fp16, 1242x375 (48x156 at stride 8), N = 253, S = 3; fp16 planes resident
                res["cfg5_fp16"] = mode_leg(WORKLOADS["cfg5"], "fp16", dev, CFG5_FRAMES, 4, breakdown=True)
                res["cfg5_fp16"]["note"] = "BASELINE configs[4] on ONE GPU (frames shard over 8 like cfg2's): fp16 feature planes and fp16 initial " \
                                           "mask logits resident, fp16 logits out; its own roofline (dominant launch class at this shape)"
                if full:
                    r32 = mode_leg(WORKLOADS["cfg5"], "fp16", dev, CFG5_FRAMES, 4, fp32_inputs=True)
                    res["cfg5_fp16"]["fp32_feature_inputs"] = {k: r32[k] for k in ("value", "ms_per_step")}
            except Exception as e:
                res["cfg5_fp16"] = {"error": repr(e)}
            try:        # SURVEY 8d's sparse-mask variant of the headline (initial mask logits - 2: ~2 % foreground)
                sp = mode_leg(wl, args.precision, dev, B, args.streams, mask_bias=-2.0)
                res["sparse_masks"] = {"value": sp["value"], "unit": "frames/s", "ms_per_step": sp["ms_per_step"], "mask_bias": -2.0,
                                       "note": "the headline step with initial mask logits N(-2, 1) (SURVEY 8d): 1-bit masks and dense MFMA "
                                               "pooling make the work independent of the mask density"}
            except Exception as e:
                res["sparse_masks"] = {"error": repr(e)}
        if world == 1 and not args.no_kernel_head:
            # SURVEY 8d's metric row a1 + a6 with the plane / mask-bit hand-off.  Primary: the cheapest pair of grades that
            # stays inside the 1e-3 contract against fp32 inputs -- KernelHead's fp16 grade (3.5e-4 .. 6.7e-4 on its outputs,
            # tests/test_gpu_parity.py) handing fp16 planes to the decode's `fp16` mode (3.9e-4 per stage); next to it the
            # fast all-bf16 pair (outside the contract, round 1's number) and the parity-grade pair (hi + lo everywhere)
            try:
                h16 = head if args.precision == "fp16" else build_head(wl, "fp16", torch.float16, dev)
                res["with_kernel_head"] = kernel_head_leg(wl, h16, "fp16", torch.float16, dev)
                res["with_kernel_head"]["precision"] = "fp16 grade in both heads: inside the 1e-3 contract against fp32 inputs"
                del h16
            except Exception as e:          # secondary leg: never lose the headline line
                res["with_kernel_head"] = {"error": repr(e)}
            for name, prec, odt in ((("fast_bf16", "bf16", torch.bfloat16), ("parity_fp32", "fp32", torch.float32)) if full else ()):
                try:
                    hk = head if args.precision == prec else build_head(wl, prec, odt, dev)
                    r = kernel_head_leg(wl, hk, prec, odt, dev)
                    res["with_kernel_head"][name] = {k: r[k] for k in ("a1_plus_a6_frames_per_s", "a1_plus_a6_ms_per_step", "a1_only_ms_per_step",
                                                                       "two_streams")}
                    del hk
                except Exception as e:
                    res["with_kernel_head"][name] = {"error": repr(e)}
            torch.cuda.empty_cache()
        if world == 1 and full and not args.no_kernel_head:
            try:
                res["panoptic_merge"] = panoptic_leg(wl, head, kplan, dev)
            except Exception as e:
                res["panoptic_merge"] = {"error": repr(e)}
        if world == 1 and full and not args.no_neck:
            # the neck and the whole head from the FPN levels: the fp16 grade (one fp16 plane of weights / activations in the
            # neck and in KernelHead, the decode's `fp16` mode: every stage inside the 1e-3 contract) first, the all-bf16 fast
            # grade (outside the contract) next to it
            try:
                res["semantic_fpn_neck"] = neck_leg(wl, "fp16", dev)
                res["semantic_fpn_neck"]["precision"] = "fp16 grade (6e-4 against the reference golden)"
                fast = neck_leg(wl, "bf16", dev)
                res["semantic_fpn_neck"]["fast_bf16"] = {k: fast[k] for k in ("frames_per_s", "ms_per_step", "mfma_TFLOPs")}
            except Exception as e:
                res["semantic_fpn_neck"] = {"error": repr(e)}
            try:
                h16 = head if args.precision == "fp16" else build_head(wl, "fp16", torch.float16, dev)
                res["full_head_from_fpn"] = full_head_leg(wl, h16, "fp16", dev)
                res["full_head_from_fpn"]["precision"] = "fp16 grade in the neck, KernelHead and the decode"
                del h16
                hk = head if args.precision == "bf16" else build_head(wl, "bf16", torch.bfloat16, dev)
                fast = full_head_leg(wl, hk, "bf16", dev)
                res["full_head_from_fpn"]["fast_bf16"] = {k: v for k, v in fast.items() if k != "note"}
                del hk
            except Exception as e:
                res.setdefault("full_head_from_fpn", {})["error"] = repr(e)
            torch.cuda.empty_cache()
        if world == 1 and full and not args.no_neck:
            try:
                res["video_cfg3"] = video_leg(dev, precision="fp16")      # fp16 grade in neck / heads, split-grade track head
                fast = video_leg(dev, precision="bf16")
                res["video_cfg3"]["fast_bf16"] = {"ms_per_frame": fast["ms_per_frame"], "eager_launches": fast["eager_launches"]}
            except Exception as e:
                res.setdefault("video_cfg3", {})["error"] = repr(e)
        if world == 1 and full and not args.no_neck:
            try:
                res["hungarian_assign"] = assign_leg(dev)
            except Exception as e:
                res["hungarian_assign"] = {"error": repr(e)}
        if world == 1 and full and not args.no_neck:
            try:
                res["train_step"] = train_leg(wl, dev, world)
            except Exception as e:
                res["train_step"] = {"error": repr(e)}
        if not full:
            res["legs_not_run"] = "secondary legs (all precision modes, panoptic merge, neck, whole head from the FPN levels, video cfg3, " \
                                  "assigner, training step) need --all-legs; profiles/r06/bench_all_legs.json holds this round's full line"
        if not args.no_cpu_baseline:                       # rank 0, at every N: the line of an N-GPU run carries it too
            res["cpu_baseline"] = cpu_baseline(wl, head, all_cores=full)
        os.write(json_fd, (json.dumps(res) + "\n").encode())
    dist.barrier()
    dist.destroy_process_group()
    rc = 0
    for w in workers:                  # ranks this process started itself (plain `python bench.py --gpus N`)
        rc |= w.wait()
    if rc:
        raise SystemExit(f"a self-launched rank exited with status {rc}")


if __name__ == "__main__":
    main()
