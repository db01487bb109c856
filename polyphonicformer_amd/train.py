"""One training step of the path: forward in training mode, the objective, backward.

What `PolyphonicFormer.forward_train` runs after `extract_feat` (polyphonic_former.py:96-129) and what mmdet's
`BaseDetector._parse_losses` + `loss.backward()` do with its result (mmdet/models/detectors/base.py:176-199):

    rpn_head.forward_train -> roi_head.forward_train -> objective = sum of the entries whose key contains 'loss' -> backward

How the work is split on the MI355X (round 5: every tensor operation of the step that touches a map or a weight is a
libpolyhead kernel, forward and backward; torch is the autograd bookkeeping between four kinds of nodes):

  * `_Rpn`: KernelHead after the neck -- three 1x1 conv + GroupNorm + ReLU towers, x = sem + loc, the static 1x1 convs,
    the hard-mask pooling (csrc/ph_train.hip products, csrc/ph_gntrain.hip GroupNorm forward / backward);
  * `_Stage` (one per KernelUpdateHead): hard-mask pooling, the whole query side -- KernelUpdator x2, attention + LN x2,
    FFN + LN x2, the fc towers, 4 MB of weights against a few hundred rows -- as ONE forward and ONE backward call
    (csrc/ph_qtrain.hip: fp32-MFMA job tables, 19 + 21 launches), the two dynamic convolutions; `feat_transform` is folded
    as in the inference path (pooling and the dynamic convolution are linear in it), gradients included;
  * `_Upsample2x`; `_Objective`: a head's or stage's losses with d(losses)/d(predictions) from the loss kernels
    (csrc/ph_loss.hip);
  * the discrete parts (Hungarian assignment on the host, targets) in between.

The result is checked against the reference's own forward + autograd backward (tests/golden/train_step.npz:
every loss, the objective and the gradient of every parameter and of the three input maps)."""
import ctypes as C

import torch

from . import _lib, engine as E, losses as Lo

LN_EPS = 1e-5
BIN_THR = 1.5 * 2.0 ** -24          # sigmoid(z) > 0.5 in fp32 (csrc/ph_common.h PH_BIN_THR)


def _gpu32(t, name):
    if not t.is_cuda:
        raise _lib.PolyheadError(f"{name} must live on the GPU: libpolyhead has no CPU path")
    return t.detach().contiguous().float()


def _param32(p, name="parameter"):
    """a parameter whose storage a kernel reads through a raw pointer: fp32, contiguous, on the GPU -- anything else would be read as
    fp32 garbage with no error (a head cast to half / bf16, a non-contiguous view), so it is refused"""
    if not p.is_cuda:
        raise _lib.PolyheadError(f"{name} must live on the GPU: libpolyhead has no CPU path")
    if p.dtype != torch.float32 or not p.is_contiguous():
        raise _lib.PolyheadError(f"training forward: fp32 contiguous parameters only ({name}: {p.dtype}, contiguous={p.is_contiguous()})")
    return p.detach()


# Test aid (None in production): `hard_mask_hook(site, logits) -> logits` is consulted wherever the training forward is about to
# BINARISE mask logits for pooling -- site = the KernelHead module or the KernelUpdateHead stage module.  The tests hand the
# reference's hard decisions in (+-1 logits) so that a comparison with the reference's gradients measures arithmetic, not a logit
# that sits within rounding of the threshold (the inference tests do the converse: the oracle follows the device's hard masks).
hard_mask_hook = None


# ---- raw calls -------------------------------------------------------------------------------------------------------------
def rows_x_map(A, X, binarize_x=False, bias=None, out=None, accumulate=False, add=None):
    """Y[b, m, p] = sum_k A[b, m, k] X[b, k, p] (+ bias[b, m]) (+ add[b, m, p]);  A [B or 1, M, K], X [B, K, *spatial] ->
    [B, M, *spatial].  `out` + accumulate: Y += (a gradient contribution added where an earlier one already lies)"""
    X = _gpu32(X, "X")
    B, K = X.shape[:2]
    HW = X[0, 0].numel()
    Ab, M, Ka = A.shape
    assert Ka == K and Ab in (1, B), (A.shape, X.shape)
    Mpad, lda = (M + 15) // 16 * 16, (K + 7) // 8 * 8
    if Mpad == M and lda == K and A.is_contiguous() and A.dtype == torch.float32:
        Ap = A.detach()
    else:
        Ap = torch.zeros((Ab, Mpad, lda), dtype=torch.float32, device=X.device)
        Ap[:, :M, :K] = A.detach()
    Y = torch.empty((B, M) + tuple(X.shape[2:]), dtype=torch.float32, device=X.device) if out is None else out
    assert Y.is_contiguous() and Y.numel() == B * M * HW and (out is not None or not accumulate) and not (accumulate and add is not None)
    if accumulate:
        add = Y
    elif add is not None:
        add = _gpu32(add, "add")
        assert add.numel() == Y.numel()
    if bias is not None:
        bias = bias.detach().contiguous().float()
        assert bias.numel() == B * M
    _lib.check(_lib.load().ph_rows_x_map_ex(_lib.ptr(Ap), Mpad * lda if Ab > 1 else 0, lda, Mpad, M, K, _lib.ptr(X), _lib.ptr(Y), B, HW,
                                            int(binarize_x), _lib.ptr(bias), _lib.ptr(add), _lib.stream_ptr()), "ph_rows_x_map")
    return Y


def map_x_mapT(G, X, binarize_g=False, out=None, rowsum=None, sum_batch=False):
    """O[b, m, k] = sum_p G[b, m, p] X[b, k, p];  G [B, M, *spatial], X [B, K, *spatial] -> [B, M, K].
    rowsum [B, M] (optional): sum_p G[b, m, p] from the same pass (binarised: the hard masks' pixel counts).
    sum_batch: summed over the images too -> [M, K] / rowsum [M] (the gradient of a static kernel and of its bias)"""
    G, X = _gpu32(G, "G"), _gpu32(X, "X")
    B, M = G.shape[:2]
    K = X.shape[1]
    HW = X[0, 0].numel()
    assert G[0, 0].numel() == HW and X.shape[0] == B
    lib = _lib.load()
    ns = lib.ph_map_x_map_t_nsplit(B, M, HW)
    part = torch.empty((B, ns, M, K), dtype=torch.float32, device=X.device)
    Bo = 1 if sum_batch else B
    if out is None:
        out = torch.empty((M, K) if sum_batch else (B, M, K), dtype=torch.float32, device=X.device)
    assert out.is_contiguous() and out.numel() == Bo * M * K
    rsp = None
    if rowsum is not None:
        assert rowsum.is_contiguous() and rowsum.numel() == Bo * M and rowsum.dtype == torch.float32
        rsp = torch.empty((B, ns, M), dtype=torch.float32, device=X.device)
    _lib.check(lib.ph_map_x_map_t_ex(_lib.ptr(G), _lib.ptr(X), _lib.ptr(part), _lib.ptr(out), B, M, K, HW, ns, int(binarize_g),
                                    _lib.ptr(rsp), _lib.ptr(rowsum), int(sum_batch), _lib.stream_ptr()), "ph_map_x_map_t")
    return out


def pool_hard_counts(m, x, dfe, pooled, cnt):
    """pooled[0] = sum_p M x, pooled[1] = sum_p M depth_feats (M = the hard masks of the logits m), cnt = pixels per mask:
    kernel_update_head.py:236-242 with feat_transform folded out (train._Stage)"""
    map_x_mapT(m, x, binarize_g=True, out=pooled[0], rowsum=cnt)
    map_x_mapT(m, dfe, binarize_g=True, out=pooled[1])


# ---- differentiable map-sized operations -------------------------------------------------------------------------------------
class _Upsample2x(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t):
        return E.upsample2x(_gpu32(t, "t"))

    @staticmethod
    def backward(ctx, g):
        g = _gpu32(g, "g")
        B, N, H2, W2 = g.shape
        out = torch.empty((B, N, H2 // 2, W2 // 2), dtype=torch.float32, device=g.device)
        _lib.check(_lib.load().ph_upsample2x_bwd(_lib.ptr(g), _lib.ptr(out), B * N, H2 // 2, W2 // 2, _lib.stream_ptr()),
                   "ph_upsample2x_bwd")
        return out


class _Objective(torch.autograd.Function):
    """The loss kernels as one differentiable scalar: forward evaluates `fn(*preds)` -> (dict of losses, dict of
    d(sum of the 'loss' entries) / d(pred)); backward hands the stored gradients on, scaled by the incoming one."""

    @staticmethod
    def forward(ctx, fn, box, *preds):
        losses, grads = fn(*[p.detach() for p in preds])
        box.update(losses)
        ctx.save_for_backward(*grads)
        total = sum(v.double() for k, v in losses.items() if "loss" in k)
        return total.float()

    @staticmethod
    def backward(ctx, g):
        return (None, None) + tuple(None if t is None else g * t for t in ctx.saved_tensors)      # no depth items: no depth gradient


upsample2x = _Upsample2x.apply


# ---- one KernelUpdateHead stage: ONE autograd node, forward and backward in libpolyhead ---------------------------------------
def _qt_names():
    """parameter names in the order of include/polyhead.h's PH_QTRAIN table: (mask branch [44], depth branch [39])"""
    def upd(u):
        return [f"{u}.dynamic_layer.weight", f"{u}.dynamic_layer.bias", f"{u}.input_layer.weight", f"{u}.input_layer.bias",
                f"{u}.input_gate.weight", f"{u}.input_gate.bias", f"{u}.update_gate.weight", f"{u}.update_gate.bias",
                f"{u}.input_norm_in.weight", f"{u}.input_norm_in.bias", f"{u}.norm_in.weight", f"{u}.norm_in.bias",
                f"{u}.norm_out.weight", f"{u}.norm_out.bias", f"{u}.input_norm_out.weight", f"{u}.input_norm_out.bias",
                f"{u}.fc_layer.weight", f"{u}.fc_layer.bias", f"{u}.fc_norm.weight", f"{u}.fc_norm.bias"]

    def rest(sfx):
        return [f"attention{sfx}.attn.in_proj_weight", f"attention{sfx}.attn.in_proj_bias", f"attention{sfx}.attn.out_proj.weight",
                f"attention{sfx}.attn.out_proj.bias", f"attention_norm{sfx}.weight", f"attention_norm{sfx}.bias",
                f"ffn{sfx}.layers.0.0.weight", f"ffn{sfx}.layers.0.0.bias", f"ffn{sfx}.layers.1.weight", f"ffn{sfx}.layers.1.bias",
                f"ffn_norm{sfx}.weight", f"ffn_norm{sfx}.bias"]
    mask = (["feat_transform.conv.weight", "feat_transform.conv.bias"] + upd("kernel_update_conv") + rest("") +
            ["mask_fcs.0.weight", "mask_fcs.1.weight", "mask_fcs.1.bias", "fc_mask.weight", "fc_mask.bias",
             "cls_fcs.0.weight", "cls_fcs.1.weight", "cls_fcs.1.bias", "fc_cls.weight", "fc_cls.bias"])
    depth = (["feat_depth_transform.conv.weight", "feat_depth_transform.conv.bias"] + upd("kernel_update_conv_depth") + rest("_depth") +
             ["depth_regs.0.weight", "depth_regs.1.weight", "depth_regs.1.bias", "fc_depth.weight", "fc_depth.bias"])
    return mask, depth


QT_MASK_NAMES, QT_DEPTH_NAMES = _qt_names()
QT_NPARAM = 44
assert len(QT_MASK_NAMES) == QT_NPARAM and len(QT_DEPTH_NAMES) == 39


def _ptr_table(tensors_mask, tensors_depth):
    """HOST array [2][44] of device pointers (the depth branch has no classification tower: 5 nulls)"""
    arr = (C.c_void_p * (2 * QT_NPARAM))()
    for i, t in enumerate(tensors_mask):
        arr[i] = t.data_ptr()
    for i, t in enumerate(tensors_depth):
        arr[QT_NPARAM + i] = t.data_ptr()
    return arr


_QT_SCRATCH = {}


def _qt_scratch(dev, n):
    """the backward's scratch (a few tens of MB): one buffer per device, reused by every stage and step -- the stages'
    backwards run one after the other on the same stream"""
    t = _QT_SCRATCH.get(dev)
    if t is None or t.numel() < n:
        t = _QT_SCRATCH[dev] = torch.empty((n,), dtype=torch.float32, device=dev)
    return t


class _Stage(torch.autograd.Function):
    """KernelUpdateHead.forward (kernel_update_head.py:212-353) in training form as ONE autograd node: hard-mask pooling of
    x / depth_feats (`ph_map_x_map_t`, feat_transform folded), the whole query side (`ph_qtrain_forward`: updators, attention,
    FFN, towers -- csrc/ph_qtrain.hip), the two dynamic convolutions (`ph_rows_x_map`); backward likewise
    (`ph_qtrain_backward` + the transposed map products).  No gradient through the hard masks (piecewise constant)."""

    @staticmethod
    def forward(ctx, meta, x, dfe, k, m, q, *params):
        lib = _lib.load()
        x, dfe, m = _gpu32(x, "x"), _gpu32(dfe, "depth_feats"), _gpu32(m, "mask_preds")
        if hard_mask_hook is not None:
            m = _gpu32(hard_mask_hook(meta["module"](), m), "mask_preds")
        k, q = _gpu32(k, "proposal_feat"), _gpu32(q, "depth_proposal")
        params = [p.detach() for p in params]
        pm, pd = params[:QT_NPARAM], params[QT_NPARAM:]
        B, N, Cc = k.shape
        L, F = meta["L"], meta["F"]
        dev, R = x.device, B * N
        pooled = torch.empty((2, B, N, Cc), dtype=torch.float32, device=dev)
        cnt = torch.empty((B, N), dtype=torch.float32, device=dev)
        pool_hard_counts(m, x, dfe, pooled, cnt)
        cls = torch.empty((B, N, L), dtype=torch.float32, device=dev)
        kern = torch.empty((2, B, N, Cc), dtype=torch.float32, device=dev)
        kbias = torch.empty((2, B, N), dtype=torch.float32, device=dev)
        obj = torch.empty((2, B, N, Cc), dtype=torch.float32, device=dev)
        saved = torch.empty((lib.ph_qtrain_saved_floats(B, N, L, F),), dtype=torch.float32, device=dev)
        ptab = _ptr_table(pm, pd)
        _lib.check(lib.ph_qtrain_forward(ptab, _lib.ptr(pooled), _lib.ptr(cnt), _lib.ptr(k), _lib.ptr(q), _lib.ptr(cls), _lib.ptr(kern),
                                         _lib.ptr(kbias), _lib.ptr(obj), _lib.ptr(saved), B, N, L, F, _lib.stream_ptr()),
                   "ph_qtrain_forward")
        mask = rows_x_map(kern[0], x, bias=kbias[0])                   # = conv(feat_transform(x), fc_mask(...)) (:317-322)
        depth = rows_x_map(kern[1], dfe, bias=kbias[1])
        ctx.meta, ctx.params = meta, params
        ctx.save_for_backward(x, dfe, m, k, q, pooled, cnt, kern, saved)
        return cls, mask, obj[0], depth, obj[1]

    @staticmethod
    def backward(ctx, gcls, gmask, gobj, gdepth, gdobj):
        lib = _lib.load()
        x, dfe, m, k, q, pooled, cnt, kern, saved = ctx.saved_tensors
        params, meta = ctx.params, ctx.meta
        pm, pd = params[:QT_NPARAM], params[QT_NPARAM:]
        B, N, Cc = k.shape
        L, F = meta["L"], meta["F"]
        dev = x.device
        gmask, gdepth = _gpu32(gmask, "grad"), _gpu32(gdepth, "grad")
        # the dynamic convolutions: d kernels (+ d bias = row sums of the map gradient), d maps
        gkern = torch.empty((2, B, N, Cc), dtype=torch.float32, device=dev)
        gkbias = torch.empty((2, B, N), dtype=torch.float32, device=dev)
        map_x_mapT(gmask, x, out=gkern[0], rowsum=gkbias[0])
        map_x_mapT(gdepth, dfe, out=gkern[1], rowsum=gkbias[1])
        need_x, need_d = ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        gx = rows_x_map(kern[0].transpose(1, 2), gmask) if need_x else None
        gd = rows_x_map(kern[1].transpose(1, 2), gdepth) if need_d else None
        gobj2 = torch.stack([_gpu32(gobj, "grad"), _gpu32(gdobj, "grad")], 0)
        # the query side
        sizes = meta["sizes"]
        gflat = torch.empty((meta["total"],), dtype=torch.float32, device=dev)
        gtab = (C.c_void_p * (2 * QT_NPARAM))()
        base = gflat.data_ptr()
        for i, off in enumerate(meta["offsets"]):
            if off >= 0:
                gtab[i] = base + 4 * off
        g_pooled = torch.empty_like(pooled)
        gk, gq = torch.empty_like(k), torch.empty_like(q)
        scratch = _qt_scratch(dev, lib.ph_qtrain_scratch_floats(B, N, L, F))
        _lib.check(lib.ph_qtrain_backward(_ptr_table(pm, pd), _lib.ptr(pooled), _lib.ptr(cnt), _lib.ptr(k), _lib.ptr(q), _lib.ptr(saved),
                                          _lib.ptr(_gpu32(gcls, "grad")), _lib.ptr(gkern), _lib.ptr(gkbias), _lib.ptr(gobj2), gtab,
                                          _lib.ptr(g_pooled), _lib.ptr(gk), _lib.ptr(gq), _lib.ptr(scratch), B, N, L, F,
                                          _lib.stream_ptr()), "ph_qtrain_backward")
        # the pooling: d map[c, p] += sum_n d pooled[n, c] M[n, p]
        if need_x:
            rows_x_map(g_pooled[0].transpose(1, 2), m, binarize_x=True, out=gx, accumulate=True)
        if need_d:
            rows_x_map(g_pooled[1].transpose(1, 2), m, binarize_x=True, out=gd, accumulate=True)
        pgrads = []
        for i, off in enumerate(meta["offsets"]):
            if off >= 0:
                pgrads.append(gflat[off:off + sizes[i]].view(meta["shapes"][i]))
        return (None, gx, gd, gk, None, gq) + tuple(pgrads)


def _stage_meta(head):
    """per-head constants of `_Stage`: parameter order, sizes and offsets of the flat gradient buffer (cached on the module)"""
    P = _lib.named_params(head)
    c = head.__dict__.get("_ph_stage_meta")
    plist = [P[n] for n in QT_MASK_NAMES] + [P[n] for n in QT_DEPTH_NAMES]
    key = tuple(id(p) for p in plist)
    if c is None or c["key"] != key:
        for p in plist:
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise _lib.PolyheadError("training forward: fp32 contiguous parameters only")
        offsets, sizes, shapes, off = [], [], [], 0
        for i in range(2 * QT_NPARAM):
            j = i if i < QT_NPARAM else i - QT_NPARAM
            have = i < QT_NPARAM or j < len(QT_DEPTH_NAMES)
            if not have:
                offsets.append(-1); sizes.append(0); shapes.append(None)
                continue
            p = plist[i if i < QT_NPARAM else QT_NPARAM + j]
            offsets.append(off); sizes.append(p.numel()); shapes.append(tuple(p.shape))
            off += (p.numel() + 3) // 4 * 4
        import weakref
        c = dict(key=key, L=head.num_classes, F=P["ffn.layers.0.0.weight"].shape[0], offsets=offsets, sizes=sizes, shapes=shapes, total=off,
                 module=weakref.ref(head))
        if P["fc_cls.weight"].shape[0] != c["L"] or P["feat_transform.conv.weight"].shape[:2] != (256, 256) or head.attention.attn.num_heads != 8:
            raise NotImplementedError("training forward: 256 channels, 8 heads, fc_cls of num_classes rows (the shipped configuration)")
        head.__dict__["_ph_stage_meta"] = c
    return c, plist


def stage_forward(head, x, dfe, k, m, q):
    """KernelUpdateHead.forward (kernel_update_head.py:212-353) in training form.  x, dfe [B, C, H, W] (gradients flow),
    k / q [B, N, C] kernels and depth kernels, m [B, N, H, W] mask logits (used through the hard mask only).
    -> cls [B, N, L], mask [B, N, H, W], obj [B, N, C], depth [B, N, H, W], dobj [B, N, C]"""
    meta, plist = _stage_meta(head)
    return _Stage.apply(meta, x, dfe, k, m, q.expand_as(k), *plist)


# ---- KernelHead after the neck: ONE autograd node -----------------------------------------------------------------------------
def gn_relu_fwd(y, gamma, beta, groups, add=None, eps=1e-5, want_out=True):
    """relu(GroupNorm(y)) of a ConvModule, fp32 NCHW (`ph_gn_train_fwd`) -> (out, out + add or None, stats [B, groups, 2])"""
    lib = _lib.load()
    B, Cc = y.shape[:2]
    HW = y[0, 0].numel()
    out = torch.empty_like(y) if (want_out or add is None) else None
    osum = torch.empty_like(y) if add is not None else None
    stats = torch.empty((B, groups, 2), dtype=torch.float32, device=y.device)
    part = torch.empty((B * groups * lib.ph_gn_train_nsplit(HW, Cc // groups) * 2,), dtype=torch.float64, device=y.device)
    _lib.check(lib.ph_gn_train_fwd(_lib.ptr(y), _lib.ptr(gamma), _lib.ptr(beta), groups, eps, _lib.ptr(add), _lib.ptr(out), _lib.ptr(osum),
                                   _lib.ptr(stats), _lib.ptr(part), B, Cc, HW, _lib.stream_ptr()), "ph_gn_train_fwd")
    return out, osum, stats


def gn_relu_bwd(y, stats, gamma, beta, groups, dyA, dyB=None):
    """backward of `gn_relu_fwd` for dy = dyA (+ dyB) -> (dx, dgamma, dbeta)"""
    lib = _lib.load()
    B, Cc = y.shape[:2]
    HW = y[0, 0].numel()
    dx = torch.empty_like(y)
    dg = torch.empty((2, Cc), dtype=torch.float32, device=y.device)
    part = torch.empty((B * Cc * lib.ph_gn_train_bwd_nsplit(HW) * 2,), dtype=torch.float64, device=y.device)
    _lib.check(lib.ph_gn_train_bwd(_lib.ptr(y), _lib.ptr(stats), _lib.ptr(gamma), _lib.ptr(beta), groups, _lib.ptr(dyA), _lib.ptr(dyB),
                                   _lib.ptr(dx), _lib.ptr(dg[0]), _lib.ptr(dg[1]), _lib.ptr(part), B, Cc, HW, _lib.stream_ptr()),
               "ph_gn_train_bwd")
    return dx, dg[0], dg[1]


RPN_NAMES = ["loc_convs.0.conv.weight", "loc_convs.0.gn.weight", "loc_convs.0.gn.bias",
             "seg_convs.0.conv.weight", "seg_convs.0.gn.weight", "seg_convs.0.gn.bias",
             "depth_convs.0.conv.weight", "depth_convs.0.gn.weight", "depth_convs.0.gn.bias",
             "init_kernels.weight", "conv_seg.weight", "conv_seg.bias", "conv_direct_depth.weight", "conv_direct_depth.bias"]


class _Rpn(torch.autograd.Function):
    """KernelHead._decode_init_proposals after the neck (kernel_head.py:245-326), training form, as ONE autograd node: three
    1x1 conv + GroupNorm + ReLU towers (`ph_rows_x_map`, `ph_gn_train_fwd`), x = sem + loc from the same pass that normalises
    sem, the static 1x1 convs with their biases in the epilogue, the hard-mask pooling; backward by hand: every second gradient
    contribution enters a kernel as an operand (no map-sized ATen add), static-kernel gradients are summed over the batch inside
    the split reduction."""

    @staticmethod
    def forward(ctx, groups, site, f0, f1, f2, *params):
        f = [_gpu32(t, "post-neck map") for t in (f0, f1, f2)]
        P = [_param32(p, n) for p, n in zip(params, RPN_NAMES)]
        W = [P[0].flatten(1), P[3].flatten(1), P[6].flatten(1)]
        B = f[0].shape[0]
        y = [rows_x_map(W[t][None], f[t]) for t in range(3)]
        loc, _, st0 = gn_relu_fwd(y[0], P[1], P[2], groups)
        sem, x, st1 = gn_relu_fwd(y[1], P[4], P[5], groups, add=loc)                   # x = sem + loc (:303)
        dfe, _, st2 = gn_relu_fwd(y[2], P[7], P[8], groups)
        W_init, W_seg, w_dd = P[9].flatten(1), P[10].flatten(1), P[12].flatten(1)
        mask_preds = rows_x_map(W_init[None], loc)                                      # :256
        seg_preds = rows_x_map(W_seg[None], sem, bias=P[11][None].expand(B, -1))        # :295
        depth_pred = rows_x_map(w_dd[None], dfe, bias=P[13][None].expand(B, -1))        # :285
        hard_src = mask_preds if hard_mask_hook is None else _gpu32(hard_mask_hook(site, mask_preds), "mask_preds")
        proposal = map_x_mapT(hard_src, x, binarize_g=True)                             # :314-320 (use_binary)
        proposal += W_init[None]                                                        # :299-300,324-326
        ctx.groups, ctx.P = groups, P
        ctx.save_for_backward(f[0], f[1], f[2], y[0], y[1], y[2], st0, st1, st2, loc, sem, dfe, x, hard_src)
        return proposal, x, mask_preds, seg_preds, dfe, depth_pred

    @staticmethod
    def backward(ctx, g_prop, g_x, g_mp, g_seg, g_dfe, g_dp):
        f0, f1, f2, y0, y1, y2, st0, st1, st2, loc, sem, dfe, x, mask_preds = ctx.saved_tensors
        P, groups = ctx.P, ctx.groups
        g_prop, g_x, g_mp, g_seg, g_dfe, g_dp = [_gpu32(t, "grad") for t in (g_prop, g_x, g_mp, g_seg, g_dfe, g_dp)]
        W_init, W_seg, w_dd = P[9].flatten(1), P[10].flatten(1), P[12].flatten(1)
        # d / d x: what arrives + the pooling's share (d pooled^T M), in one pass; sem and loc both receive it
        gxs = rows_x_map(g_prop.transpose(1, 2), mask_preds, binarize_x=True, add=g_x)
        dy = [rows_x_map(W_init.t()[None], g_mp), rows_x_map(W_seg.t()[None], g_seg), rows_x_map(w_dd.t()[None], g_dp)]
        gy0, dg0, db0 = gn_relu_bwd(y0, st0, P[1], P[2], groups, dy[0], gxs)
        gy1, dg1, db1 = gn_relu_bwd(y1, st1, P[4], P[5], groups, dy[1], gxs)
        gy2, dg2, db2 = gn_relu_bwd(y2, st2, P[7], P[8], groups, dy[2], g_dfe)
        gW_init = map_x_mapT(g_mp, loc, sum_batch=True)
        gW_init += g_prop.sum(0)
        L = W_seg.shape[0]
        gb_seg = torch.empty((L,), dtype=torch.float32, device=x.device)
        gW_seg = map_x_mapT(g_seg, sem, rowsum=gb_seg, sum_batch=True)
        gb_dd = torch.empty((1,), dtype=torch.float32, device=x.device)
        gw_dd = map_x_mapT(g_dp, dfe, rowsum=gb_dd, sum_batch=True)
        gf, gW = [], []
        for t, (gy, ft) in enumerate(((gy0, f0), (gy1, f1), (gy2, f2))):
            gf.append(rows_x_map(P[3 * t].flatten(1).t()[None], gy) if ctx.needs_input_grad[2 + t] else None)
            gW.append(map_x_mapT(gy, ft, sum_batch=True).view(P[3 * t].shape))
        return (None, None, gf[0], gf[1], gf[2], gW[0], dg0, db0, gW[1], dg1, db1, gW[2], dg2, db2, gW_init.view(P[9].shape),
                gW_seg.view(P[10].shape), gb_seg, gw_dd.view(P[12].shape), gb_dd)


def rpn_forward(head, feats):
    """KernelHead._decode_init_proposals after the neck (kernel_head.py:245-336), training form (no stuff rows)"""
    P = _lib.named_params(head)
    plist = [P[n] for n in RPN_NAMES]
    groups = head.norm_cfg.get("num_groups", 32)
    proposal, x, mask_preds, seg_preds, dfe, depth_pred = _Rpn.apply(groups, head, feats[0], feats[1], feats[2], *plist)
    B = x.shape[0]
    depth_proposal = P["conv_direct_depth.weight"].flatten(1)[None].expand(B, 1, -1)           # :286-289
    return dict(proposal=proposal, x=x, mask_preds=mask_preds, seg_preds=seg_preds, dfe=dfe, depth_proposal=depth_proposal,
                depth_pred=depth_pred)


# ---- SemanticFPNWrapper in training: the neck's forward as differentiable libpolyhead nodes (round 5) ------------------------------
class _Conv3x3(torch.autograd.Function):
    """F.conv2d(X, W [M, K, 3, 3], stride, padding 1) on `ph_conv3x3_train` (nine shifted rows-x-map products); backward: the input
    gradient is the same kernel on flipped / transposed taps (stride 2: its transposed-stride mode), the weight gradient nine
    shifted map x map^T products summed over the batch (`ph_conv3x3_wgrad`)"""

    @staticmethod
    def _taps(W, transpose, flip):
        M, K = W.shape[:2]
        t = torch.empty((9, K, M) if transpose else (9, M, K), dtype=torch.float32, device=W.device)
        _lib.check(_lib.load().ph_conv3x3_taps(_lib.ptr(W), _lib.ptr(t), M, K, int(transpose), int(flip), 0, _lib.stream_ptr()), "ph_conv3x3_taps")
        return t

    @staticmethod
    def forward(ctx, X, W, stride):
        X, Wd = _gpu32(X, "X"), _gpu32(W, "W")
        B, K, Hi, Wi = X.shape
        M = Wd.shape[0]
        Ho, Wo = (Hi - 1) // stride + 1, (Wi - 1) // stride + 1
        Y = torch.empty((B, M, Ho, Wo), dtype=torch.float32, device=X.device)
        _lib.check(_lib.load().ph_conv3x3_train(_lib.ptr(_Conv3x3._taps(Wd, False, False)), M, K, _lib.ptr(X), _lib.ptr(Y), B, Hi, Wi, Ho, Wo,
                                                stride, 0, _lib.stream_ptr()), "ph_conv3x3_train")
        ctx.stride = stride
        ctx.save_for_backward(X, Wd)
        return Y

    @staticmethod
    def backward(ctx, gY):
        X, W = ctx.saved_tensors
        lib, s = _lib.load(), ctx.stride
        gY = _gpu32(gY, "grad")
        B, K, Hi, Wi = X.shape
        M, Ho, Wo = W.shape[0], gY.shape[2], gY.shape[3]
        gX = gW = None
        if ctx.needs_input_grad[0]:
            gX = torch.empty_like(X)
            if s == 1:
                _lib.check(lib.ph_conv3x3_train(_lib.ptr(_Conv3x3._taps(W, True, True)), K, M, _lib.ptr(gY), _lib.ptr(gX), B, Ho, Wo, Hi, Wi, 1, 0,
                                                _lib.stream_ptr()), "ph_conv3x3_train(dgrad)")
            else:
                _lib.check(lib.ph_conv3x3_train(_lib.ptr(_Conv3x3._taps(W, True, False)), K, M, _lib.ptr(gY), _lib.ptr(gX), B, Ho, Wo, Hi, Wi, 2, 1,
                                                _lib.stream_ptr()), "ph_conv3x3_train(dgrad, stride 2)")
        if ctx.needs_input_grad[1]:
            ns = lib.ph_map_x_map_t_nsplit(B, M, Ho * Wo)
            part = torch.empty((B, ns, M, K), dtype=torch.float32, device=X.device)
            taps = torch.empty((9, M, K), dtype=torch.float32, device=X.device)
            _lib.check(lib.ph_conv3x3_wgrad(_lib.ptr(gY), _lib.ptr(X), _lib.ptr(part), _lib.ptr(taps), B, M, K, Hi, Wi, Ho, Wo, s, ns,
                                            _lib.stream_ptr()), "ph_conv3x3_wgrad")
            gW = torch.empty_like(W)
            _lib.check(lib.ph_conv3x3_taps(_lib.ptr(gW), _lib.ptr(taps), M, K, 0, 0, 1, _lib.stream_ptr()), "ph_conv3x3_taps")
        return gX, gW, None


class _GNReLU(torch.autograd.Function):
    """relu(GroupNorm(y)) (+ add: the running sum over the neck's towers, semantic_fpn.py:217-220) -- csrc/ph_gntrain.hip"""

    @staticmethod
    def forward(ctx, y, gamma, beta, groups, add):
        y = _gpu32(y, "y")
        g, b = _param32(gamma, "GroupNorm weight"), _param32(beta, "GroupNorm bias")
        out, osum, stats = gn_relu_fwd(y, g, b, groups, add=None if add is None else _gpu32(add, "add"), want_out=False)
        ctx.groups, ctx.has_add = groups, add is not None
        ctx.save_for_backward(y, stats, g, b)
        return out if add is None else osum

    @staticmethod
    def backward(ctx, gout):
        y, stats, g, b = ctx.saved_tensors
        gout = _gpu32(gout, "grad")
        dx, dg, db = gn_relu_bwd(y, stats, g, b, ctx.groups, gout)
        return dx, dg, db, None, (gout if ctx.has_add else None)


class _NeckOuts(torch.autograd.Function):
    """conv_pred + the aux convs (semantic_fpn.py:152-178, 222-229): 1x1 conv + GroupNorm + ReLU each, on the tower sum; backward
    in one node so that the three input-gradient contributions land in ONE buffer (the second and third as the product's add source)"""

    @staticmethod
    def forward(ctx, groups, s, *params):
        s = _gpu32(s, "tower sum")
        P = [_param32(p, "neck output conv parameter") for p in params]
        n = len(P) // 3
        ys, sts, outs = [], [], []
        for j in range(n):
            y = rows_x_map(P[3 * j].flatten(1)[None], s)
            o, _, st = gn_relu_fwd(y, P[3 * j + 1], P[3 * j + 2], groups)
            ys.append(y); sts.append(st); outs.append(o)
        ctx.groups, ctx.P, ctx.n = groups, P, n
        ctx.save_for_backward(s, *ys, *sts)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        sv = ctx.saved_tensors
        P, n, s = ctx.P, ctx.n, sv[0]
        ys, sts = sv[1:1 + n], sv[1 + n:1 + 2 * n]
        gs, grads = None, []
        for j in range(n):
            gy, dg, db = gn_relu_bwd(ys[j], sts[j], P[3 * j + 1], P[3 * j + 2], ctx.groups, _gpu32(gouts[j], "grad"))
            if ctx.needs_input_grad[1]:
                W = P[3 * j].flatten(1)
                gs = rows_x_map(W.t()[None], gy) if gs is None else rows_x_map(W.t()[None], gy, out=gs, accumulate=True)
            grads += [map_x_mapT(gy, s, sum_batch=True).view(P[3 * j].shape), dg, db]
        return (None, gs) + tuple(grads)


def neck_forward_train(neck, inputs):
    """SemanticFPNWrapper.forward (funcs/semantic_fpn.py:198-235) for the shipped configuration, every operation a differentiable
    libpolyhead node: per level 3x3 conv (level 0: stride 2) + GroupNorm + ReLU, x2 bilinear upsamples in between, the towers'
    outputs summed inside the last GroupNorm pass of each level, conv_pred / aux convs.  fp32 NCHW, the reference's arithmetic
    (products on hi + lo bf16 MFMA like the heads' map products).  Gradients flow to every parameter and to the four FPN inputs."""
    G = neck.groups
    total = None
    for lvl, level in enumerate(neck.convs_all_levels):
        t = inputs[lvl].float()
        if lvl == neck.cat_coors_level and neck.pos_cfg is not None:
            t = t + neck._posenc(t.shape[-2], t.shape[-1], t.device)[None]                    # semantic_fpn.py:202-209
        for j in range(level.n):
            m = getattr(level, f"conv{j}")
            y = _Conv3x3.apply(t, m.conv.weight, m.stride)
            last = j == level.n - 1
            t = _GNReLU.apply(y, m.gn.weight, m.gn.bias, G, total if (last and total is not None) else None)
            if not last:                              # levels 2, 3: an x2 upsample follows every conv but the last (:117-150)
                t = upsample2x(t)
        total = t
    outs = [neck.conv_pred] + list(neck.aux_convs)
    plist = []
    for m in outs:
        plist += [m.conv.weight, m.gn.weight, m.gn.bias]
    res = _NeckOuts.apply(G, total, *plist)
    return list(res)


# ---- the two heads' training forwards (ONE implementation: the API methods and TrainStep both call these) --------------------
def parse_losses(losses):
    """mmdet BaseDetector._parse_losses (base.py:188-199): the objective is the sum of the entries with 'loss' in the key"""
    return sum(v.mean() for k, v in losses.items() if "loss" in k and torch.is_tensor(v))


def _attach(values, total):
    """values: {name: detached scalar}, total: the differentiable objective these values add up to (their 'loss' entries,
    already weighted).  Returns the dict the reference's forward_train returns, with every 'loss' entry carrying an equal
    share of `total`'s graph, so that what mmdet forms from it -- `sum(the 'loss' entries)`, base.py:188-199 -- has exactly
    the gradient of the objective, through ONE autograd node per head and stage.  (The loss kernels return
    d(sum of a head's losses) / d(prediction), not one gradient per entry: only the SUM of the entries is differentiable
    in a meaningful way, which is all the reference's runner ever differentiates.)"""
    keys = [k for k in values if "loss" in k]
    share = (total - total.detach()) / max(len(keys), 1)
    return {k: (v + share if "loss" in k else v) for k, v in values.items()}


def check_stage_topology(head):
    """the assumptions stage_forward hard-codes (the shipped configuration, configs/_base_/models/polyphonic_former.py:111-165);
    the constructors reject everything else, this guards modules altered after construction"""
    bad = []
    if len(head.cls_fcs) != 3 or len(head.mask_fcs) != 3 or len(head.depth_regs) != 2: bad.append("num_cls_fcs / num_mask_fcs != 1")
    if head.dropout != 0.0: bad.append("dropout")
    if head.conv_kernel_size != 1 or head.mask_transform_stride != 1 or head.feat_gather_stride != 1: bad.append("kernel size / strides")
    if head.hard_mask_thr != 0.5: bad.append("hard_mask_thr")
    if not head.with_ffn or len(head.ffn.layers) != 3 or len(head.ffn.layers[0]) != 3: bad.append("FFN layout")   # (Linear, act, Dropout), Linear, Dropout
    if bad:
        raise NotImplementedError("training forward: unsupported KernelUpdateHead topology: " + ", ".join(bad))


def check_rpn_topology(head):
    bad = []
    if len(head.loc_convs) != 1 or len(head.seg_convs) != 1 or len(head.depth_convs) != 1: bad.append("num_*_convs != 1")
    if head.conv_kernel_size != 1 or not head.use_binary or not head.proposal_feats_with_obj: bad.append("kernel size / use_binary")
    if head.feat_downsample_stride > 1 and head.feat_refine: bad.append("feat_refine")
    if head.feat_downsample_stride not in (1, 2): bad.append("feat_downsample_stride")
    if bad:
        raise NotImplementedError("training forward: unsupported KernelHead topology: " + ", ".join(bad))


def _valid_pixels(gt_masks_i, gt_sem_seg_i):
    """1.0 where any instance or stuff mask of the image is set (kernel_head.py:415, kernel_update.py:238:
    `torch.cat((gt_masks, gt_sem_seg)).sum(0).bool()`), without materialising the concatenation"""
    v = None
    for t in (gt_masks_i, gt_sem_seg_i):
        if t.shape[0]:
            a = t.ne(0).any(dim=0)
            v = a if v is None else (v | a)
    if v is None:
        return torch.zeros(gt_masks_i.shape[1:], dtype=torch.float32, device=gt_masks_i.device)
    return v.float()


def _fast_assign_ok(assigner, sampler):
    """the batched descriptor path covers the shipped training configuration: one-to-one Hungarian matching on class / mask / dice
    costs (DepthCost weight 0, polyphonic_former.py:170-192) and the pseudo sampler"""
    from .assigner import _MaskAssignerBase, MaskPseudoSampler
    return (isinstance(assigner, _MaskAssignerBase) and assigner.topk == 1 and isinstance(sampler, MaskPseudoSampler)
            and (assigner.depth_cost is None or assigner.depth_cost.weight == 0))


def _step_gt(cache, hard, gt_masks, gt_labels, gt_sem_seg, gt_sem_cls, gt_depth):
    if cache is None:
        cache = {}
    if hard not in cache:
        cache[hard] = Lo.StepGT(gt_masks, gt_labels, gt_sem_seg, gt_sem_cls, gt_depth, hard)
    return cache[hard]


def rpn_forward_train(h, feats, img_metas, gt_masks, gt_labels, gt_sem_seg, gt_sem_cls, gt_depth, want_grads=False, gt_cache=None):
    """KernelHead.forward_train, kernel_head.py:349-454, on the three post-neck maps (gradients flow into `feats` when they
    require them).  Returns (losses, r): `losses` = the reference's dict with the 'loss' entries attached to the graph
    (`_attach`), `depth_dense` logged only (base.py:198 leaves it out of the objective); r = the differentiable training-
    mode decode (no stuff rows).  want_grads: losses['_grads'] = d(sum of the 'loss' entries) / d(scaled mask, seg and
    direct depth predictions)."""
    check_rpn_topology(h)
    if h.assigner is None:
        raise ValueError("forward_train needs train_cfg (assigner / sampler)")
    gt_raw = gt_masks
    if h.hard_target:                     # local to the rpn side, as in the reference (kernel_head.py:400-403)
        gt_masks = [m.bool().float() for m in gt_masks]
    r = rpn_forward(h, feats)
    up = (lambda t: upsample2x(t)) if h.feat_downsample_stride == 2 else (lambda t: t)
    smask, sseg, sdep0 = up(r["mask_preds"]), up(r["seg_preds"]), up(r["depth_pred"])         # :364-398
    N = h.num_proposals + h.num_stuff_classes
    if _fast_assign_ok(h.assigner, h.sampler):
        # round 5: batched assignment (one pixel pass + one D2H for all images), targets as pointer tables, ONE loss call
        gt = _step_gt(gt_cache, bool(h.hard_target), gt_raw, gt_labels, gt_sem_seg, gt_sem_cls, gt_depth)
        assigns = Lo.assign_batch(h.assigner, smask.detach(), None, gt)
        desc = Lo.build_desc(h, gt, assigns, h.num_proposals, h.train_cfg, roi=False)
        kept = {}

        def fn(mp, sp, dp):
            ls, g = Lo.fused_losses(h, desc, mp, None, dp, sp, with_grads=True)
            kept.update(mask_pred=g["mask_pred"], seg_preds=g["seg_preds"], depth_pred=g["depth_pred"])
            return ls, (g["mask_pred"], g["seg_preds"], g["depth_pred"])

        values = {}
        total = _Objective.apply(fn, values, smask, sseg, sdep0)
        losses = _attach(values, total)
        if gt_depth is not None:
            losses["depth_dense"] = Lo.dense_depth_loss(h, sdep0.detach(), gt_depth)          # :438-442, logged only
        if want_grads:
            losses["_grads"] = kept
        return losses, r
    sdep = sdep0.detach().expand(-1, N, -1, -1)
    srs = []
    for i in range(len(img_metas)):                                                           # :411-426
        valid = _valid_pixels(gt_masks[i], gt_sem_seg[i])
        ar = h.assigner.assign(smask[i].detach(), None, gt_masks[i], gt_labels[i], img_metas[i], depth_pred=sdep[i],
                               gt_depth=gt_depth[i], gt_valid=valid)
        sr = h.sampler.sample(ar, smask[i].detach(), gt_masks[i], depth=sdep[i])
        sr.valid_mask = valid
        srs.append(sr)
    targets = h.get_targets(srs, gt_masks, h.train_cfg, True, gt_sem_seg=gt_sem_seg, gt_sem_cls=gt_sem_cls, gt_depth=gt_depth)
    kept = {}

    def fn(mp, sp, dp):
        ls, g = Lo.rpn_losses(h, mp, sp, dp.expand(-1, N, -1, -1), *targets, with_grads=True)
        kept.update(mask_pred=g["mask_pred"], seg_preds=g["seg_preds"], depth_pred=g["depth_pred"])
        return ls, (g["mask_pred"], g["seg_preds"], g["depth_pred"])

    values = {}
    total = _Objective.apply(fn, values, smask, sseg, sdep0)
    losses = _attach(values, total)
    if gt_depth is not None:
        losses["depth_dense"] = Lo.dense_depth_loss(h, sdep0.detach(), gt_depth)              # :438-442, logged only
    if want_grads:
        losses["_grads"] = kept
    return losses, r


def rpn_outputs(h, r):
    """what KernelHead.forward_train hands to the roi head besides the losses (kernel_head.py:444-454): the stuff rows
    appended when cat_stuff_mask, all tensors still on the graph"""
    B = r["x"].shape[0]
    nt, L = h.num_thing_classes, h.num_classes
    mask_preds, k, q = r["mask_preds"], r["proposal"], r["depth_proposal"]
    if h.cat_stuff_mask:
        mask_preds = torch.cat([mask_preds, r["seg_preds"][:, nt:L]], dim=1)
        stuff = _lib.named_params(h)["conv_seg.weight"][nt:L].flatten(1)
        k = torch.cat([k, stuff[None].expand(B, -1, -1)], dim=1)
    N = k.shape[1]
    return k, mask_preds, q.expand(B, N, -1)


def roi_forward_train(ih, x, dfe, k, mask_preds, q, depth_pred, img_metas, gt_masks, gt_labels, gt_sem_seg, gt_sem_cls, gt_depth,
                      want_grads=False, gt_cache=None):
    """KernelUpdateIterHead.forward_train, kernel_update.py:159-280.  x, dfe [B, C, H, W]; k / q [B, N, C] kernels and depth
    kernels; mask_preds [B, N, H, W]; depth_pred [B, 1, H, W].  Every stage: forward in training form, the Hungarian
    assignment on the previous stage's detached predictions, pseudo sampling, targets, the stage's losses as one autograd
    node.  Returns (losses, last): losses = {s{stage}_{name}} weighted and attached (`_attach`), last = the final stage's
    (object_feats, cls_score, mask_preds, scaled_mask_preds)."""
    if ih.post_assign:
        raise NotImplementedError                       # as the reference (:222-223)
    if not ih.mask_assigner:
        raise ValueError("forward_train needs train_cfg (assigner / sampler per stage)")
    B, N = k.shape[:2]
    up = ih.mask_head[0].mask_upsample_stride
    if up not in (1, 2):
        raise NotImplementedError("libpolyhead: mask_upsample_stride must be 1 or 2")
    scale = (lambda t: upsample2x(t)) if up == 2 else (lambda t: t)
    prev_mask = scale(mask_preds.detach()).detach()                                           # :179-191
    if all(_fast_assign_ok(a, sm) for a, sm in zip(ih.mask_assigner, ih.mask_sampler)):
        return _roi_forward_train_fast(ih, x, dfe, k, mask_preds, q, prev_mask, scale, gt_masks, gt_labels, gt_sem_seg, gt_sem_cls, gt_depth,
                                       want_grads, gt_cache)
    prev_depth = scale(depth_pred.detach().expand(-1, N, -1, -1).contiguous()).detach()
    prev_cls = [None] * B                                                                      # :193-196
    if ih.hard_target:
        gt_masks = [m.bool().float() for m in gt_masks]
    # the pixels some ground-truth mask covers (:238): the same for every stage -- evaluated once per image instead of once per
    # image and stage (a 24 MB concatenation + reduction each at the assign stride of cfg2)
    valids = [_valid_pixels(gt_masks[i], gt_sem_seg[i]) for i in range(B)]
    total, m, assign, values, grads = 0.0, mask_preds, [], {}, []
    cls = smask = None
    for s in range(ih.num_stages):
        head = ih.mask_head[s]
        check_stage_topology(head)
        cls, m, k, depth, q = stage_forward(head, x, dfe, k, m, q)
        smask, sdepth = scale(m), scale(depth)                                                 # training: every stage (:131)
        srs = []
        if s < ih.assign_stages:
            assign = []
        for i in range(B):
            valid = valids[i]
            if s < ih.assign_stages:
                c = None if prev_cls[i] is None else prev_cls[i][:ih.num_proposals, :ih.num_thing_classes]
                assign.append(ih.mask_assigner[s].assign(prev_mask[i][:ih.num_proposals], c, gt_masks[i], gt_labels[i], img_metas[i],
                                                         depth_pred=prev_depth[i][:ih.num_proposals], gt_depth=gt_depth[i],
                                                         gt_valid=valid))
            sr = ih.mask_sampler[s].sample(assign[i], smask[i].detach(), gt_masks[i], depth=sdepth[i].detach())
            sr.valid_mask = valid
            srs.append(sr)
        targets = head.get_targets(srs, gt_masks, gt_labels, ih.train_cfg[s], True, gt_sem_seg=gt_sem_seg, gt_sem_cls=gt_sem_cls,
                                   gt_depth=gt_depth)
        kept = {}

        def fn(cs, mp, dp, head=head, targets=targets, kept=kept):
            ls, g = Lo.stage_losses(head, cs, mp, dp, *targets, with_grads=True)
            kept.update(g)
            return ls, (g["cls_score"], g["mask_pred"], g["depth_pred"])

        box = {}
        w = ih.stage_loss_weights[s]
        total = total + w * _Objective.apply(fn, box, cls, smask, sdepth)
        for key, v in box.items():
            values[f"s{s}_{key}"] = v * w
        grads.append(kept)
        prev_mask, prev_cls, prev_depth = smask.detach(), cls.detach(), sdepth.detach()       # :273-276
    losses = _attach(values, total)
    if want_grads:
        losses["_grads"] = grads
    return losses, (k, cls, m, smask)


def _roi_forward_train_fast(ih, x, dfe, k, mask_preds, q, prev_mask, scale, gt_masks, gt_labels, gt_sem_seg, gt_sem_cls, gt_depth, want_grads,
                            gt_cache):
    """the stage loop of `roi_forward_train` on the batched / descriptor path (round 5): per stage ONE `_Stage` node, two
    upsamples, one batched assignment (`ph_match_sums` over all images + one D2H + the host Hungarian solves), one pointer-table
    upload and ONE loss call (`ph_train_losses`).  No sampler gathers, no materialised targets: every target row is a row of the
    step's ground truth (`losses.StepGT`).  The previous stage's depth predictions only feed the DepthCost, whose weight is 0
    on this path, so they are not formed."""
    B = k.shape[0]
    Np, nt = ih.num_proposals, ih.num_thing_classes
    gt = _step_gt(gt_cache, bool(ih.hard_target), gt_masks, gt_labels, gt_sem_seg, gt_sem_cls, gt_depth)
    total, m, values, grads, assigns, prev_cls = 0.0, mask_preds, {}, [], None, None
    cls = smask = None
    for s in range(ih.num_stages):
        head = ih.mask_head[s]
        check_stage_topology(head)
        cls, m, k, depth, q = stage_forward(head, x, dfe, k, m, q)
        smask, sdepth = scale(m), scale(depth)                                                 # training: every stage (:131)
        if s < ih.assign_stages:
            c = None if prev_cls is None else prev_cls[:, :Np, :nt]
            assigns = Lo.assign_batch(ih.mask_assigner[s], prev_mask[:, :Np], c, gt)           # :231-251
        desc = Lo.build_desc(head, gt, assigns, Np, ih.train_cfg[s], roi=True)
        kept = {}

        def fn(cs, mp, dp, head=head, desc=desc, kept=kept):
            ls, g = Lo.fused_losses(head, desc, mp, cs, dp, None, with_grads=True)
            kept.update(g)
            return ls, (g["cls_score"], g["mask_pred"], g["depth_pred"])

        box = {}
        w = ih.stage_loss_weights[s]
        obj = _Objective.apply(fn, box, cls, smask, sdepth)
        total = total + (obj if w == 1 else w * obj)
        for key, v in box.items():
            values[f"s{s}_{key}"] = v if w == 1 else v * w
        grads.append(kept)
        prev_mask, prev_cls = smask.detach(), cls.detach()                                     # :273-276
    losses = _attach(values, total)
    if want_grads:
        losses["_grads"] = grads
    return losses, (k, cls, m, smask)


# ---- the step ------------------------------------------------------------------------------------------------------------------
class TrainStep:
    """rpn_head: KernelHead, roi_head: KernelUpdateIterHead, both built with train_cfg.  `forward_backward` evaluates one
    step on the three post-neck maps and leaves `.grad` on every parameter of the two heads (accumulating, like autograd; averaged over the ranks
    when torch.distributed runs with more than one) and returns (losses, objective, gradients of the three maps).  The
    same two functions back the reference-named API methods (`KernelHead.forward_train`,
    `KernelUpdateIterHead.forward_train`), whose loss dicts mmdet's `_parse_losses` + `backward()` consume unchanged; this
    class adds the bucketed gradient all-reduce.  Use as a context manager (or call `close()`) to take the gradient hooks
    off the parameters again."""

    def __init__(self, rpn_head, roi_head, bucket_bytes=32 << 20, group=None):
        self.rpn, self.roi = rpn_head, roi_head
        if rpn_head.assigner is None or not roi_head.mask_assigner:
            raise ValueError("TrainStep needs heads built with train_cfg (assigner / sampler)")
        check_rpn_topology(rpn_head)
        for h in roi_head.mask_head:
            check_stage_topology(h)
        # data parallel: one process per GPU, gradients averaged by bucketed all-reduces (RCCL over xGMI) that start while
        # backward is still running (dist.GradBuckets); on one rank nothing is sent
        from .dist import GradBuckets
        self.group = group
        self.buckets = GradBuckets(self.parameters(), bucket_bytes=bucket_bytes, group=group)

    def close(self):
        self.buckets.remove()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def parameters(self):
        return [p for n, p in _lib.named_params(self.rpn).items() if not n.startswith("localization_fpn.")] + list(_lib.named_params(self.roi).values())

    def forward_backward(self, feats, img_metas, gt_masks, gt_labels, gt_sem_seg, gt_sem_cls, gt_depth, backward=True):
        feats = [f.detach().float().contiguous().requires_grad_(True) for f in feats]
        for f in feats:
            if not f.is_cuda:
                raise _lib.PolyheadError("the post-neck maps must live on the GPU: libpolyhead has no CPU path")
        with torch.enable_grad(), Lo.reduce_group(self.group):
            gt_cache = {}                                   # the step's ground truth (losses.StepGT), shared by the two heads
            rpn_losses, r = rpn_forward_train(self.rpn, feats, img_metas, gt_masks, gt_labels, gt_sem_seg, gt_sem_cls, gt_depth,
                                              gt_cache=gt_cache)
            k, mask_preds, q = rpn_outputs(self.rpn, r)
            losses, _ = roi_forward_train(self.roi, r["x"], r["dfe"], k, mask_preds, q, r["depth_pred"], img_metas, gt_masks, gt_labels,
                                          gt_sem_seg, gt_sem_cls, gt_depth, gt_cache=gt_cache)
            losses.update(rpn_losses)                       # polyphonic_former.py:126
            total = parse_losses(losses)
            if backward:
                self.buckets.start()
                total.backward()
                self.buckets.finish()
        return {k_: v.detach() for k_, v in losses.items()}, total.detach(), [f.grad for f in feats]
