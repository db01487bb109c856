"""Host orchestration of the decode hot path on one GPU: owns the device buffers (torch tensors are
used for allocation and streams only) and issues the libpolyhead kernels on the current stream.

Per frame-batch the launch sequence is

    ingest(x), ingest(depth_feats), binarize(mask_preds)
    for s in 0..S-1:   pool -> query_stage(pre, post) -> dynconv (bits for s < S-1, logits for s = S-1)
    upsample2x(mask) ; dynconv(depth) ; upsample2x(depth)

No step synchronises or allocates once a `DecodePlan` exists, so the whole sequence can be captured
into a HIP graph (`DecodePlan.capture`)."""
import ctypes as C

import torch

from . import _lib
from .pack import pack_stage

import collections

# Precision modes of the decode path.  One row = the arithmetic of every operator of a stage:
#   feat  : element format / planes of the feature maps (ingest, pool)        query : the query-side GEMMs
#   conv  : the dynamic 1x1 conv (feature planes x kernel planes)             kern_fmt : what the query kernel emits for it
Mode = collections.namedtuple("Mode", "name feat query conv kern_fmt FP KP feat_dtype")
import os as _os
# query side of the modes that emit fp16 dynamic kernels: hybrid grade (updator half hi/lo bf16, attention / FFN / tower half one
# fp16 plane, include/polyhead.h PH_PREC_QHYBRID) unless PH_QUERY_FULL_SPLIT=1 asks for hi/lo bf16 throughout
_QH = _lib.PH_PREC_SPLIT if _os.environ.get("PH_QUERY_FULL_SPLIT") else _lib.PH_PREC_QHYBRID
MODES = {
    # fast: bf16 everywhere, one plane (6e-3 per stage against fp32)
    "bf16": Mode("bf16", _lib.PH_PREC_BF16, _lib.PH_PREC_BF16, _lib.PH_PREC_BF16, _lib.PH_KERN_BF16_PLANES, 1, 1, torch.bfloat16),
    # bf16 feature planes as they are (cfg2's input dtype), everything computed FROM them at fp32 grade: exact 0/1 x bf16
    # pooling, hi/lo split query GEMMs, hi/lo dynamic kernels x one feature plane (<= 1e-3 on identical inputs)
    "mixed": Mode("mixed", _lib.PH_PREC_BF16, _lib.PH_PREC_SPLIT, _lib.PH_PREC_BF16_KSPLIT, _lib.PH_KERN_BF16_PLANES, 1, 2, torch.bfloat16),
    # as `mixed`, but the dynamic kernels as ONE fp16 plane: the conv converts its bf16 feature fragments to fp16 in registers
    # (exact) and runs one f16 MFMA -- the single-plane conv speed (kernel rounding 2^-12: 2.5e-4 per stage on its own)
    "mixed16": Mode("mixed16", _lib.PH_PREC_BF16, _QH, _lib.PH_PREC_BF16_KF16, _lib.PH_KERN_F16, 1, 1, torch.bfloat16),
    # fp16 feature planes / kernels / outputs (cfg5), fp32-grade query side
    "fp16": Mode("fp16", _lib.PH_PREC_F16, _QH, _lib.PH_PREC_F16, _lib.PH_KERN_F16, 1, 1, torch.float16),
    # parity grade: every operand hi + lo (1.6e-5 per stage)
    "fp32": Mode("fp32", _lib.PH_PREC_SPLIT, _lib.PH_PREC_SPLIT, _lib.PH_PREC_SPLIT, _lib.PH_KERN_BF16_PLANES, 2, 2, None),
}
MODES["split"] = MODES["fp32"]
# arithmetic of the kernels that know two grades only (KernelHead, the neck, the track head): fast or fp32 grade
PREC = {"bf16": _lib.PH_PREC_BF16, "split": _lib.PH_PREC_SPLIT, "fp32": _lib.PH_PREC_SPLIT,
        "mixed": _lib.PH_PREC_SPLIT, "mixed16": _lib.PH_PREC_SPLIT, "fp16": _lib.PH_PREC_SPLIT}
# KernelHead's post-neck part (a1) and the neck have a third grade: "fp16" = ONE fp16 plane of the maps and weights (2^-12 per operand, one
# MFMA per product), whose planes and mask bits the decode's `fp16` mode adopts as they are
KHEAD_PREC = dict(PREC, fp16=_lib.PH_PREC_F16)
OUT_CODE = {torch.float32: _lib.PH_OUT_F32, torch.bfloat16: _lib.PH_OUT_BF16, torch.float16: _lib.PH_OUT_F16}


def mode_of(p):
    """Mode from a Mode, a mode name, or a legacy precision code (PH_PREC_BF16 / PH_PREC_SPLIT)"""
    if isinstance(p, Mode):
        return p
    if isinstance(p, str):
        return MODES[p]
    return {_lib.PH_PREC_BF16: MODES["bf16"], _lib.PH_PREC_SPLIT: MODES["fp32"], _lib.PH_PREC_F16: MODES["fp16"]}[p]


def hw_padded(hw):
    return (hw + 127) // 128 * 128


def n_padded(n):
    return (n + 31) // 32 * 32


def _require_gpu(t, name):
    if not t.is_cuda:
        raise _lib.PolyheadError(f"{name} must live on the GPU: libpolyhead has no CPU path")


class StagePack:
    """device-resident packed weights of one KernelUpdateHead stage"""

    def __init__(self, sd, prefix, num_classes, prec, device):
        wb, wf, lay = pack_stage(sd, prefix, num_classes, prec)
        self.wb = wb.to(device)
        self.wf = wf.to(device)
        self.lay = lay
        self.num_classes = num_classes
        self.prec = prec


def plan_env_key():
    """the environment switches a DecodePlan's kernel choice reads at construction: part of the module API's plan-cache key"""
    return tuple(_os.environ.get(k) for k in ("PH_POOL_NSPLIT", "PH_CONV_UP2", "PH_CONV_POOLX", "PH_POOLX_NSPLIT", "PH_UP2_SHARED_WGS"))


def default_nsplit(B, HW, frame_invariant=False):
    """pixel ranges per frame for the split-K pooling: 4*B*nsplit workgroups should fill the chip's resident
    slots (2 workgroups per CU x 256 CUs) without a partial second generation.  `frame_invariant`: the split of a ONE-frame
    launch whatever B is -- the split fixes the order in which a frame's partial sums are added, so this is what keeps a frame's
    bits independent of the frames that share its launch (the module API; throughput callers -- bench.py -- split by B)"""
    import os
    if os.environ.get("PH_POOL_NSPLIT"):
        return int(os.environ["PH_POOL_NSPLIT"])
    nchunks = hw_padded(HW) // 128
    ns = max(1, 512 // (4 * (1 if frame_invariant else B)))
    return int(min(ns, 32, max(1, nchunks)))


# ---- thin op wrappers (each = one C-ABI call) ---------------------------------------------------
def ingest(x, prec, out=None):
    """fp32 [B,256,H,W] -> bf16 planes int16 [P,B,256,HWp]"""
    _require_gpu(x, "x")
    B, Cc, H, W = x.shape
    if Cc != 256:
        raise _lib.PolyheadError("libpolyhead supports 256 channels (the shipped configs)")
    x = x.contiguous().float()
    P = 2 if prec == _lib.PH_PREC_SPLIT else 1          # PH_PREC_BF16 / PH_PREC_F16: one plane
    if out is None:
        out = torch.empty((P, B, 256, hw_padded(H * W)), dtype=torch.int16, device=x.device)
    lib = _lib.load()
    _lib.check(lib.ph_ingest_features(_lib.ptr(x), _lib.ptr(out), B, H * W, prec, _lib.stream_ptr()), "ph_ingest_features")
    return out


def binarize(m, out=None):
    """mask logits [B,N,H,W] (fp32, or fp16 / bf16 as a 16-bit KernelHead grade hands them over) -> mask bits int32 [B,Npad,HWp/32]"""
    _require_gpu(m, "mask_preds")
    B, N, H, W = m.shape
    m = m.contiguous() if m.dtype in OUT_CODE else m.contiguous().float()
    if out is None:
        out = torch.empty((B, n_padded(N), hw_padded(H * W) // 32), dtype=torch.int32, device=m.device)
    lib = _lib.load()
    _lib.check(lib.ph_binarize_if(_lib.ptr(m), OUT_CODE[m.dtype], 0, _lib.ptr(out), B, N, H * W, None, _lib.stream_ptr()), "ph_binarize")
    return out


def pool(xp, dp, bits, N, HW, prec, nsplit=None, out=None, counts=None):
    """`counts`: optional int32 [B, nsplit, Npad] that receives the masks' pixel counts per pixel range (ph_pool_counts)"""
    B = xp.shape[1]
    if nsplit is None:
        nsplit = default_nsplit(B, HW)
    if out is None:
        out = torch.empty((B, nsplit, n_padded(N), 512), dtype=torch.float32, device=xp.device)
    lib = _lib.load()
    if counts is not None:
        _lib.check(lib.ph_pool_counts(_lib.ptr(xp), _lib.ptr(dp), _lib.ptr(bits), _lib.ptr(out), _lib.ptr(counts), B, N, HW, nsplit, prec,
                                      _lib.stream_ptr()), "ph_pool_counts")
    else:
        _lib.check(lib.ph_pool(_lib.ptr(xp), _lib.ptr(dp), _lib.ptr(bits), _lib.ptr(out), B, N, HW, nsplit, prec,
                               _lib.stream_ptr()), "ph_pool")
    return out


def query_stage(partial, bits, k_in, q_in, pack, N, HW, cls_sigmoid=False, outs=None, workspace=None, phases=3,
                kern_fmt=_lib.PH_KERN_BF16_PLANES, counts=None):
    B, nsplit = partial.shape[0], partial.shape[1]
    dev = partial.device
    prec = pack.prec
    P = 1 if kern_fmt == _lib.PH_KERN_F16 else (2 if prec == _lib.PH_PREC_SPLIT else 1)     # planes of `kern`
    Npad = n_padded(N)
    lib = _lib.load()
    if outs is None:
        outs = dict(obj=torch.empty((B, N, 256), dtype=torch.float32, device=dev),
                    dobj=torch.empty((B, N, 256), dtype=torch.float32, device=dev),
                    cls=torch.empty((B, N, pack.num_classes), dtype=torch.float32, device=dev),
                    kern=torch.empty((P, 2, B, Npad, 256), dtype=torch.int16, device=dev),
                    kbias=torch.empty((2, B, Npad), dtype=torch.float32, device=dev))
    if workspace is None:
        workspace = torch.empty((lib.ph_query_workspace_bytes(B, N, prec),), dtype=torch.uint8, device=dev)
    tail = (_lib.ptr(k_in), _lib.ptr(q_in), _lib.ptr(pack.wb), _lib.ptr(pack.wf), C.byref(pack.lay),
            _lib.ptr(outs["obj"]), _lib.ptr(outs["dobj"]), _lib.ptr(outs["cls"]), 1 if cls_sigmoid else 0, _lib.ptr(outs["kern"]),
            _lib.ptr(outs["kbias"]), _lib.ptr(workspace), workspace.numel(), B, N, HW, prec, kern_fmt, phases, _lib.stream_ptr())
    if counts is not None:
        _lib.check(lib.ph_query_stage_counts(_lib.ptr(partial), nsplit, _lib.ptr(bits), _lib.ptr(counts), *tail), "ph_query_stage_counts")
    else:
        _lib.check(lib.ph_query_stage(_lib.ptr(partial), nsplit, _lib.ptr(bits), *tail), "ph_query_stage")
    return outs


def dynconv(planes, kern, kbias, branch, N, HW, prec, bits_out=None, logits_out=None, out_dtype=_lib.PH_OUT_F32):
    """per-frame dynamic kernels: kern [P,2,B,Npad,256], kbias [2,B,Npad]; `branch` = mask (0) / depth (1)."""
    B, Npad = kern.shape[2], kern.shape[3]
    lib = _lib.load()
    kptr = C.c_void_p(kern.data_ptr() + branch * B * Npad * 256 * 2)
    bptr = C.c_void_p(kbias.data_ptr() + branch * B * Npad * 4)
    _lib.check(lib.ph_dynconv(_lib.ptr(planes), kptr, 2 * B * Npad * 256, Npad * 256, bptr, Npad, _lib.ptr(bits_out),
                              _lib.ptr(logits_out), out_dtype, N * HW, B, N, HW, prec, _lib.stream_ptr()), "ph_dynconv")
    return bits_out if bits_out is not None else logits_out


def dynconv_poolx(planes, kern, kbias, N, HW, prec, bits_out, partial):
    """non-final stage's mask conv (branch 0) -> mask bits, AND the next stage's pooling of the x map (columns 0 .. 255 of
    `partial` [B, nsplit, Npad, 512]) from one read of the plane (ph_dynconv_poolx); kern [1,2,B,Npad,256], kbias [2,B,Npad]"""
    B, Npad = kern.shape[2], kern.shape[3]
    lib = _lib.load()
    _lib.check(lib.ph_dynconv_poolx(_lib.ptr(planes), _lib.ptr(kern), Npad * 256, _lib.ptr(kbias), Npad, _lib.ptr(bits_out), _lib.ptr(partial),
                                    partial.shape[1], B, N, HW, prec, _lib.stream_ptr()), "ph_dynconv_poolx")
    return bits_out


def pool_depth_only(dp, bits, N, HW, prec, partial, counts):
    """the depth_feats half of a pooling whose x half ph_dynconv_poolx has written: columns 256 .. 511 of `partial` and the pixel counts"""
    B, nsplit = partial.shape[0], partial.shape[1]
    lib = _lib.load()
    _lib.check(lib.ph_pool_counts(_lib.ptr(dp), None, _lib.ptr(bits), C.c_void_p(partial.data_ptr() + 256 * 4), _lib.ptr(counts), B, N, HW, nsplit, prec,
                                  _lib.stream_ptr()), "ph_pool_counts(depth)")


def dynconv_up2(planes, kern, kbias, branch, N, H, W, prec, up_out, logits_out=None, out_dtype=_lib.PH_OUT_F16, workgroups=0):
    """final-stage dynamic conv + x2 bilinear upsample in one kernel (ph_dynconv_up2): kern [1,2,B,Npad,256] (one 16-bit
    plane), kbias [2,B,Npad]; writes up_out [B,N,2H,2W] and, when given, the low-resolution logits [B,N,H,W].
    `workgroups`: 0 = one per CU; a launch that shares the GPU ends sooner with 1.5 per CU (ph_dynconv_up2_wgs; same values)"""
    B, Npad = kern.shape[2], kern.shape[3]
    lib = _lib.load()
    kptr = C.c_void_p(kern.data_ptr() + branch * B * Npad * 256 * 2)
    bptr = C.c_void_p(kbias.data_ptr() + branch * B * Npad * 4)
    _lib.check(lib.ph_dynconv_up2_wgs(_lib.ptr(planes), kptr, Npad * 256, bptr, Npad, _lib.ptr(logits_out), _lib.ptr(up_out), out_dtype,
                                      B, N, H, W, prec, int(workgroups), _lib.stream_ptr()), "ph_dynconv_up2")
    return up_out


def static_conv(planes, wplanes, bias, N, HW, prec, logits_out, out_rows):
    """the same 1x1 conv weights for every frame: wplanes [P,Npad,256] bf16 planes, bias fp32 [Npad];
    writes fp32 logits into rows [0, N) of each frame of `logits_out` ([B, out_rows, H, W] view)."""
    B = planes.shape[1]
    Npad = wplanes.shape[1]
    lib = _lib.load()
    _lib.check(lib.ph_dynconv(_lib.ptr(planes), _lib.ptr(wplanes), Npad * 256, 0, _lib.ptr(bias), 0, None,
                              _lib.ptr(logits_out), _lib.PH_OUT_F32, out_rows * HW, B, N, HW, prec, _lib.stream_ptr()),
               "ph_dynconv(static)")
    return logits_out


def upsample2x(src, out=None):
    """[..., H, W] fp32 or bf16 -> [..., 2H, 2W] (bilinear, align_corners=False)"""
    _require_gpu(src, "src")
    src = src.contiguous()
    H, W = src.shape[-2:]
    planes = src.numel() // (H * W)
    if out is None:
        out = torch.empty(tuple(src.shape[:-2]) + (2 * H, 2 * W), dtype=src.dtype, device=src.device)
    if src.dtype not in OUT_CODE:
        raise _lib.PolyheadError("upsample2x: fp32, bf16 or fp16 only")
    dt = OUT_CODE[src.dtype]
    lib = _lib.load()
    _lib.check(lib.ph_upsample2x(_lib.ptr(src), _lib.ptr(out), dt, planes, H, W, _lib.stream_ptr()), "ph_upsample2x")
    return out


# ---- the S-stage plan ------------------------------------------------------------------------------
class DecodePlan:
    """All buffers for `simple_test_mask_preds` at one (B, N, H, W, precision, output dtype)."""

    def __init__(self, packs, B, N, H, W, prec, out_dtype=torch.float32, device="cuda:0", nsplit=None, frame_invariant=False):
        """`frame_invariant` (round 6): every choice that touches a frame's arithmetic -- the pooling's pixel split, the fused / two-
        kernel final stage -- is the ONE-frame launch's at any B, so a frame's outputs do not depend on its batch (the module API's
        default; a clip's frames through one launch equal the per-frame loop bit for bit)"""
        self.packs, self.S = packs, len(packs)
        self.frame_invariant = frame_invariant
        self.B, self.N, self.H, self.W, self.HW = B, N, H, W, H * W
        self.mode = mode_of(prec)
        self.prec, self.out_dtype = self.mode.feat, out_dtype           # `prec`: the feature planes' code (ingest / pool)
        if any(p.prec != self.mode.query for p in packs):
            raise _lib.PolyheadError(f"stage packs are not packed for mode '{self.mode.name}'")
        self.nsplit = nsplit or default_nsplit(B, self.HW, frame_invariant)
        dev = torch.device(device)
        P, KP = self.mode.FP, self.mode.KP
        Npad, HWp = n_padded(N), hw_padded(self.HW)
        L = packs[0].num_classes
        e = lambda shape, dt: torch.empty(shape, dtype=dt, device=dev)
        # static inputs (graph-capturable)
        self.x = e((B, 256, H, W), torch.float32)
        self.dfe = e((B, 256, H, W), torch.float32)
        self.k0 = e((B, N, 256), torch.float32)
        self.q0 = e((B, N, 256), torch.float32)
        self.m0 = e((B, N, H, W), torch.float32)
        # internals
        # (zero-filled when the planes carry pixel padding: 16-bit inputs are copied into them row by row and the kernels read whole
        # 128-pixel chunks -- a NaN bit pattern in the padding would survive the multiplication with a zero mask bit)
        z = (lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)) if HWp != self.HW else e
        self.xp = z((P, B, 256, HWp), torch.int16)
        self.dp = z((P, B, 256, HWp), torch.int16)
        self.bits = e((B, Npad, HWp // 32), torch.int32)
        self.partial = e((B, self.nsplit, Npad, 512), torch.float32)
        self.pcount = e((B, self.nsplit, Npad), torch.int32)      # the masks' pixel counts per pixel range, pool -> query kernel
        self.ws = e((_lib.load().ph_query_workspace_bytes(B, N, self.mode.query),), torch.uint8)
        self.stage_out = [dict(obj=e((B, N, 256), torch.float32), dobj=e((B, N, 256), torch.float32),
                               cls=e((B, N, L), torch.float32), kern=e((KP, 2, B, Npad, 256), torch.int16),
                               kbias=e((2, B, Npad), torch.float32)) for _ in range(self.S)]
        # outputs
        self.mask = e((B, N, H, W), out_dtype)
        self.depth = e((B, N, H, W), out_dtype)
        self.mask_up = e((B, N, 2 * H, 2 * W), out_dtype)
        self.depth_up = e((B, N, 2 * H, 2 * W), out_dtype)
        self.graph = None
        self.handoff_runs = 0          # runs that started from another kernel's planes (tests observe the path taken)
        # Final stage: conv + x2 upsample in ONE kernel where it exists (W = 256, one-plane conv grades, 16-bit outputs): the
        # low-resolution logits are not written and re-read, the depth branch's are not written at all (no caller of
        # simple_test / simple_test_mask_preds ever receives them: kernel_update.py:338-345,401).  `want_depth_lowres` makes the
        # fused depth launch write them too.
        # Chosen when the batch has at least two image rows per CU (B * H >= 512): below that a workgroup's range is one or two rows
        # and the silent halo row above it doubles its work -- one frame per launch: 34 us against 26 us for the two kernels, equal
        # at 2-4 frames, ahead from 8 (same box).  PH_CONV_UP2=1 forces the fused form at any size (tests), =0 the two-kernel form.
        _up2 = _os.environ.get("PH_CONV_UP2", "auto")
        self.fused_up = (KP == 1 and _up2 != "0" and (_up2 == "1" or (1 if frame_invariant else B) * H >= 512)
                         and bool(_lib.load().ph_dynconv_up2_supported(N, H, W, self.mode.conv, OUT_CODE[out_dtype])))
        self.want_depth_lowres = False
        # workgroups of the fused final stage when the plan is one part of a multi-stream step: 1.5 per CU (PH_UP2_SHARED_WGS=n; stages())
        self.up2_shared_wgs = int(_os.environ.get("PH_UP2_SHARED_WGS") or
                                  (3 * torch.cuda.get_device_properties(device).multi_processor_count // 2 if torch.device(device).type == "cuda" else 0))
        # Round 6: a non-final stage's mask conv also pools the x map for the NEXT stage from the same read of the plane
        # (ph_dynconv_poolx), the next stage then pools depth_feats alone: 33.5 MB instead of 50 MB per frame and stage boundary at
        # cfg2.  One workgroup per (frame, pixel range) and CU: for launches that fill the chip, at least 16 tiles of 64 pixels per
        # workgroup (its prologue loads the frame's kernels, its epilogue writes Npad x 256 sums).  Throughput plans split by B
        # (256 / B ranges).  `frame_invariant` plans: the kernel's pixel ranges are k_pool's, and in the bf16 / fp16 grades its sums are
        # k_pool's bit for bit at the same split (tests/test_gpu_kernels.py) -- with the plan's one-frame split the choice between the two
        # forms is invisible in a frame's outputs and may follow B; `mixed16` pools the fp16-converted tile (1e-6 apart): its invariant
        # plans keep the separate kernels.  PH_CONV_POOLX=0 / 1: never / wherever supported
        px_env = _os.environ.get("PH_CONV_POOLX", "auto")
        if frame_invariant:
            self.nsplit_px = self.nsplit
            px_ok = self.mode.conv in (_lib.PH_PREC_BF16, _lib.PH_PREC_F16) and HWp // (64 * max(self.nsplit_px, 1)) >= 8
        else:
            self.nsplit_px = int(_os.environ.get("PH_POOLX_NSPLIT") or max(1, min(256 // max(B, 1), HWp // (64 * 16))))
            px_ok = True
        self.poolx = (KP == 1 and self.S > 1 and px_env != "0" and px_ok and (px_env == "1" or B * self.nsplit_px >= 192)
                      and bool(_lib.load().ph_dynconv_poolx_supported(N, self.mode.conv)))
        if self.poolx:
            self.partial_px = e((B, self.nsplit_px, Npad, 512), torch.float32)
            self.pcount_px = e((B, self.nsplit_px, Npad), torch.int32)

    @property
    def out_code(self):
        return OUT_CODE[self.out_dtype]

    def renew_outputs(self):
        """Give the next `run` fresh output tensors (allocation only, no copy).  The reference's methods return tensors the
        caller owns (SURVEY 8b "Threading / ownership"): the module API calls this before every run, so that results kept
        from an earlier call -- e.g. the key frame's while the reference frame is decoded,
        polyphonic_former_video.py:208-242 -- are never overwritten.  Captured plans (bench) keep their fixed buffers."""
        if self.graph is not None:
            raise _lib.PolyheadError("renew_outputs on a captured plan")
        e = lambda t: torch.empty_like(t)
        self.mask, self.depth, self.mask_up, self.depth_up = e(self.mask), e(self.depth), e(self.mask_up), e(self.depth_up)
        last = self.stage_out[-1]
        for k in ("obj", "dobj", "cls"):
            last[k] = e(last[k])

    def set_inputs(self, x, dfe, k0, q0, m0):
        """x / dfe: fp32 NCHW (converted to planes by the ingest kernel inside `run`), or 16-bit NCHW tensors of the
        mode's own plane format (bf16 for 'bf16' / 'mixed', fp16 for 'fp16'), which ARE the plane format when H*W is
        a multiple of 128: they are adopted as they are and no ingest pass runs.  Other sizes (cfg5: 48 x 156): the rows are copied
        into the planes' first H*W pixels, the padding up to the next multiple of 128 stays zero (round 6)."""
        self.feat_is_bf16 = x.dtype in (torch.bfloat16, torch.float16) and dfe.dtype == x.dtype     # 16-bit plane inputs
        if self.feat_is_bf16:
            if self.mode.feat_dtype != x.dtype:
                raise _lib.PolyheadError(f"{x.dtype} feature inputs need a mode with that plane format (bf16: 'bf16' / "
                                         f"'mixed' / 'mixed16', fp16: 'fp16'; this plan: '{self.mode.name}')")
            self.xp.view(x.dtype)[0, :, :, :self.HW].copy_(x.reshape(self.B, 256, self.HW))
            self.dp.view(x.dtype)[0, :, :, :self.HW].copy_(dfe.reshape(self.B, 256, self.HW))
        else:
            self.x.copy_(x)
            self.dfe.copy_(dfe)
        self.k0.copy_(k0.reshape(self.B, self.N, 256))
        self.q0.copy_(q0.reshape(self.B, self.N, 256))   # materialises the stride-0 expand view
        if m0.dtype in OUT_CODE and self.m0.dtype != m0.dtype:
            # mask logits are binarised from the format they arrive in (16-bit: what a 16-bit KernelHead grade hands over and what
            # cfg2's bf16 inputs mean -- half the bytes; never a rounding of fp32 logits).  A captured graph holds the old buffer:
            # it is dropped, the owner captures again
            self.m0 = torch.empty_like(self.m0, dtype=m0.dtype)
            self.graph = None
        self.m0.copy_(m0)

    def ingest(self):
        if not getattr(self, "feat_is_bf16", False):
            ingest(self.x, self.prec, out=self.xp)
            ingest(self.dfe, self.prec, out=self.dp)
        binarize(self.m0, out=self.bits)

    def stages(self, xp=None, dp=None):
        """the S stages on the plan's own planes, or on read-only planes another kernel produced; the mask bits are
        always the plan's own (they are rewritten by every non-final stage)"""
        xp = self.xp if xp is None else xp
        dp = self.dp if dp is None else dp
        k, q = self.k0, self.q0
        for s in range(self.S):
            last = s == self.S - 1
            if getattr(self, "debug_bits", None) is not None:      # tests: the hard masks stage s pools with (eager runs only)
                self.debug_bits.append(self.bits.clone())
            if s > 0 and self.poolx:
                # the x map's sums came with the previous stage's conv (same read of the plane): depth_feats alone here
                pool_depth_only(dp, self.bits, self.N, self.HW, self.prec, self.partial_px, self.pcount_px)
                part, cnt = self.partial_px, self.pcount_px
            else:
                pool(xp, dp, self.bits, self.N, self.HW, self.prec, self.nsplit, out=self.partial, counts=self.pcount)
                part, cnt = self.partial, self.pcount
            if s == 0 and getattr(self, "on_first_pool", None) is not None:
                self.on_first_pool()       # multi-part callers skew their parts by one phase (an event recorded here)
            o = query_stage(part, self.bits, k, q, self.packs[s], self.N, self.HW, cls_sigmoid=last,
                            outs=self.stage_out[s], workspace=self.ws, kern_fmt=self.mode.kern_fmt, counts=cnt,
                            phases=3 | (_lib.PH_QUERY_WIDE if getattr(self, "shares_gpu", False) else 0))
            cv = self.mode.conv
            if not last:
                if self.poolx:
                    dynconv_poolx(xp, o["kern"], o["kbias"], self.N, self.HW, cv, self.bits, self.partial_px)
                else:
                    dynconv(xp, o["kern"], o["kbias"], 0, self.N, self.HW, cv, bits_out=self.bits)
            else:
                # each x2 upsample directly behind the conv that wrote its source (240 MB of logits at cfg2, 24 frames): on
                # its own a part's four launches take 659 us in this order against 823 us as conv, conv, up, up; inside the
                # four-stream step the other parts' streams evict the logits either way (same-box A/B: no difference)
                if self.fused_up:
                    # a part of a multi-stream step (`shares_gpu`): 1.5 workgroups per CU -- the other parts' query kernels hold CUs when
                    # this launch starts, and what cannot start at once leaves a shorter tail (+0.9 % on the step, profiles/r06/knob_sweep.txt)
                    wg = self.up2_shared_wgs if (getattr(self, "shares_gpu", False) and self.B * self.H >= 4 * self.up2_shared_wgs) else 0
                    dynconv_up2(xp, o["kern"], o["kbias"], 0, self.N, self.H, self.W, cv, self.mask_up, logits_out=self.mask,
                                out_dtype=self.out_code, workgroups=wg)
                    dynconv_up2(dp, o["kern"], o["kbias"], 1, self.N, self.H, self.W, cv, self.depth_up,
                                logits_out=self.depth if self.want_depth_lowres else None, out_dtype=self.out_code, workgroups=wg)
                else:
                    dynconv(xp, o["kern"], o["kbias"], 0, self.N, self.HW, cv, logits_out=self.mask, out_dtype=self.out_code)
                    upsample2x(self.mask, out=self.mask_up)
                    dynconv(dp, o["kern"], o["kbias"], 1, self.N, self.HW, cv, logits_out=self.depth, out_dtype=self.out_code)
                    upsample2x(self.depth, out=self.depth_up)
            k, q = o["obj"], o["dobj"]

    def run(self):
        """one pass: ingest + S stages + final upsample, on the current stream"""
        self.ingest()
        self.stages()

    def run_from_planes(self, xp, dp, bits, k0, q0):
        """same, starting from feature planes / mask bits another kernel already produced (KernelHead hand-off): no
        ingest pass.  The planes are only read; the bits are copied (0.6 MB per frame at cfg2) because the stages
        rewrite them, so the producer's tensors stay valid for its caller."""
        self.handoff_runs += 1
        self.bits.copy_(bits)
        self.k0.copy_(k0.reshape(self.B, self.N, 256))
        self.q0.copy_(q0.reshape(self.B, self.N, 256))
        self.stages(xp, dp)

    def capture(self):
        """record `run` into a HIP graph (replay with `replay`)"""
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self.run()      # warm-up outside capture (lazy module load, attribute setup)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.run()
        return self.graph

    def replay(self):
        self.graph.replay()

    def outputs(self):
        last = self.stage_out[-1]
        depth = self.depth if (not self.fused_up or self.want_depth_lowres) else None      # fused final stage: never written
        return dict(obj=last["obj"], dobj=last["dobj"], cls=last["cls"], mask=self.mask, depth=depth,
                    mask_up=self.mask_up, depth_up=self.depth_up)


# ---- KernelHead (a1) ---------------------------------------------------------------------------------
def _planes_of(w64, P, fp16=False):
    """float64 [..] -> int16 [P, ...] bf16 hi(/lo) planes, or ONE fp16 plane"""
    w = w64.to(torch.float32)
    if fp16:
        return w.to(torch.float16).view(torch.int16)[None].contiguous()
    hi = w.to(torch.bfloat16)
    out = [hi.view(torch.int16)]
    if P == 2:
        out.append((w - hi.float()).to(torch.bfloat16).view(torch.int16))
    return torch.stack(out, 0).contiguous()


def _pad_rows32(w):
    r = (-w.shape[0]) % 32
    return torch.cat([w, w.new_zeros((r,) + tuple(w.shape[1:]))], 0) if r else w


class KernelHeadPack:
    """device-resident packed parameters of KernelHead's post-neck part (kernel_head.py:142-211)"""

    def __init__(self, sd, prec, device, groups):
        P = 2 if prec == _lib.PH_PREC_SPLIT else 1
        h = prec == _lib.PH_PREC_F16                                         # one fp16 plane of every weight
        g = lambda k: sd[k].detach().to("cpu", torch.float64)
        convs = torch.stack([g(f"{n}_convs.0.conv.weight").reshape(256, 256) for n in ("loc", "seg", "depth")], 0)
        self.wplanes = _planes_of(convs, P, h).to(device)                    # [P,3,256,256]
        self.gn = torch.stack([torch.stack([g(f"{n}_convs.0.gn.weight"), g(f"{n}_convs.0.gn.bias")], 0)
                               for n in ("loc", "seg", "depth")], 0).float().contiguous().to(device)   # [3,2,256]
        w_init = g("init_kernels.weight").reshape(-1, 256)
        w_seg = g("conv_seg.weight").reshape(-1, 256)
        w_dd = g("conv_direct_depth.weight").reshape(1, 256)
        self.n_init, self.n_seg = w_init.shape[0], w_seg.shape[0]
        self.init_planes = _planes_of(_pad_rows32(w_init), P, h).to(device)
        self.seg_planes = _planes_of(_pad_rows32(w_seg), P, h).to(device)
        self.dd_planes = _planes_of(_pad_rows32(w_dd), P, h).to(device)
        z = lambda n: torch.zeros(n, dtype=torch.float32)
        sb = z(self.seg_planes.shape[1]); sb[:self.n_seg] = g("conv_seg.bias").float()
        self.seg_bias = sb.to(device)
        db = z(32); db[0] = float(g("conv_direct_depth.bias")[0])
        self.dd_bias = db.to(device)
        # the same three weights as MFMA A fragments for the fused second GEMM of ph_khead_fused
        from .pack import pack_b32
        frag = lambda pl: torch.stack([pack_b32(pl[p].cpu()) for p in range(P)], 0).contiguous().to(device)
        self.init_frag, self.seg_frag, self.dd_frag = frag(self.init_planes), frag(self.seg_planes), frag(self.dd_planes)
        # the three conv weights in the same fragment form: the A operand of ph_khead_onepass (one-plane grades)
        self.conv_frag = torch.stack([pack_b32(self.wplanes[0, m].cpu()) for m in range(3)], 0).contiguous().to(device) if P == 1 else None
        self.w_init_f32 = w_init.float().contiguous().to(device)
        self.w_seg_f32 = w_seg.float().contiguous().to(device)
        self.w_dd_f32 = sd["conv_direct_depth.weight"].detach().float().contiguous().to(device)   # [1,256,1,1]
        self.prec, self.groups = prec, groups


class KernelHeadPlan:
    """buffers + launch sequence of KernelHead's post-neck part for one (B, H, W)."""

    def __init__(self, pack, B, H, W, num_thing_classes, num_classes, cat_stuff, device, want_f32=True, nsplit=None,
                 logit_dtype=torch.float32, onepass=None, frame_invariant=False):
        """`onepass`: None = ph_khead_onepass whenever the geometry / grade allows it (and PH_KHEAD_TWOPASS is unset),
        False = always the two-pass ph_khead_fused.  `logit_dtype`: fp32 (the reference API) or fp16 (one-pass form only)
        for mask_preds / seg_preds / depth_pred."""
        self.pack, self.B, self.H, self.W, self.HW = pack, B, H, W, H * W
        self.n_thing_cls, self.n_cls, self.cat_stuff = num_thing_classes, num_classes, cat_stuff
        self.Nq = pack.n_init
        self.n_stuff = (num_classes - num_thing_classes) if cat_stuff else 0
        self.N = self.Nq + self.n_stuff
        prec = pack.prec
        P = 2 if prec == _lib.PH_PREC_SPLIT else 1
        HWp = hw_padded(self.HW)
        dev = torch.device(device)
        e = lambda shape, dt: torch.empty(shape, dtype=dt, device=dev)
        self.f = [None, None, None]        # set_inputs: borrowed from the caller or allocated on first copy
        self._borrowed = set()
        self.in_planes = False
        self.xp, self.dp = e((P, B, 256, HWp), torch.int16), e((P, B, 256, HWp), torch.int16)
        self.x_f32 = e((B, 256, H, W), torch.float32) if want_f32 else None
        self.dfe_f32 = e((B, 256, H, W), torch.float32) if want_f32 else None
        import os
        lib = _lib.load()
        self.ws1 = None
        self.onepass = False
        if onepass is not False and not os.environ.get("PH_KHEAD_TWOPASS") and pack.conv_frag is not None:
            # the input format is only known at set_inputs: the fp32 form has the stricter condition (HW % 4 == 0)
            self.onepass = bool(lib.ph_khead_onepass_supported(B, self.HW, pack.groups, prec, _lib.PH_IN_F32_NCHW))
        if onepass and not self.onepass:
            raise _lib.PolyheadError("ph_khead_onepass does not support this geometry / grade")
        if logit_dtype != torch.float32 and not self.onepass:
            raise _lib.PolyheadError("16-bit KernelHead logits need the one-pass form")
        self.logit_dtype = logit_dtype
        if self.onepass:
            # hand-off state of the persistent launch; zeroed ONCE (its last 256 bytes are the sticky time-out words, which the
            # calls never clear)
            self.ws1 = torch.zeros((lib.ph_khead_onepass_workspace_bytes(B, self.HW),), dtype=torch.uint8, device=dev)
        self.mask_preds = e((B, self.N, H, W), logit_dtype)
        self.seg_preds = e((B, pack.n_seg, H, W), logit_dtype)
        self.depth_pred = e((B, 1, H, W), logit_dtype)
        self.bits = e((B, n_padded(self.N), HWp // 32), torch.int32)
        self.nsplit = nsplit or default_nsplit(B, self.HW, frame_invariant)     # (see DecodePlan: the one-frame split at any B)
        self.partial = e((B, self.nsplit, n_padded(self.Nq), 512), torch.float32)
        self.proposal = e((B, self.N, 256), torch.float32)
        # the two-pass kernels' workspace: the path itself, or the in-call fallback of a one-pass launch that gave up
        self.ws = e((lib.ph_khead_workspace_bytes(B, self.HW, pack.groups),), torch.uint8)
        self.w_stuff = pack.w_seg_f32[num_thing_classes:num_classes].contiguous() if self.n_stuff else None

    def renew_outputs(self):
        """fresh tensors for everything `KernelHead.simple_test_rpn` hands to its caller (the 9-tuple and the plane / bit
        hand-off to KernelUpdateIterHead), see DecodePlan.renew_outputs"""
        e = lambda t: None if t is None else torch.empty_like(t)
        self.xp, self.dp, self.bits = e(self.xp), e(self.dp), e(self.bits)
        self.x_f32, self.dfe_f32 = e(self.x_f32), e(self.dfe_f32)
        self.mask_preds, self.seg_preds, self.depth_pred = e(self.mask_preds), e(self.seg_preds), e(self.depth_pred)
        self.proposal = e(self.proposal)

    def set_inputs(self, feats):
        """the three post-neck maps: contiguous fp32 device tensors of the plan's shape are used where they are (the
        kernels only read them; a captured graph keeps pointing at them, so they stay referenced here), anything
        else is copied into the plan's own buffers"""
        self.in_planes = feats[0].dtype == torch.int16
        if self.in_planes:          # bf16 planes [P][B][256][HWp] from the neck (SemanticFPNWrapper.forward_planes)
            P = 2 if self.pack.prec == _lib.PH_PREC_SPLIT else 1
            for i, src in enumerate(feats):
                if tuple(src.shape) != (P, self.B, 256, hw_padded(self.HW)) or not src.is_contiguous():
                    raise _lib.PolyheadError("plane inputs must be contiguous int16 [P][B][256][HWp]")
                self.f[i] = src
            self._borrowed = {t.data_ptr() for t in self.f}
            return
        for i, src in enumerate(feats):
            if (src.dtype == torch.float32 and src.is_contiguous() and src.device == self.xp.device
                    and tuple(src.shape) == (self.B, 256, self.H, self.W)):
                self.f[i] = src.detach()
            else:
                if self.f[i] is None or self.f[i].data_ptr() in self._borrowed:
                    self.f[i] = torch.empty((self.B, 256, self.H, self.W), dtype=torch.float32, device=self.xp.device)
                self.f[i].copy_(src)
        self._borrowed = {t.data_ptr() for t, src in zip(self.f, feats) if t.data_ptr() == src.data_ptr()}

    def run(self):
        lib, pk, s = _lib.load(), self.pack, _lib.stream_ptr
        B, HW, prec = self.B, self.HW, pk.prec
        fmt = _lib.PH_IN_PLANES if self.in_planes else _lib.PH_IN_F32_NCHW
        if self.onepass:
            # one read of the three maps: conv1x1+GN+ReLU x3, x = sem + loc, the static 1x1 convs AND the mask bits
            # (kernel_head.py:250-331, :314-317) in one persistent launch; the object pooling reads the thing rows of
            # the full bit tensor in place
            _lib.check(lib.ph_khead_onepass(_lib.ptr(self.f[0]), _lib.ptr(self.f[1]), _lib.ptr(self.f[2]), _lib.ptr(pk.conv_frag),
                                            _lib.ptr(pk.gn), pk.groups, 1e-5, _lib.ptr(pk.init_frag), self.Nq,
                                            _lib.ptr(pk.seg_frag), _lib.ptr(pk.seg_bias), pk.n_seg, _lib.ptr(pk.dd_frag),
                                            _lib.ptr(pk.dd_bias), self.n_thing_cls, self.n_stuff, _lib.ptr(self.xp), _lib.ptr(self.dp),
                                            _lib.ptr(self.x_f32), _lib.ptr(self.dfe_f32), _lib.ptr(self.mask_preds),
                                            _lib.ptr(self.seg_preds), _lib.ptr(self.depth_pred), OUT_CODE[self.logit_dtype],
                                            _lib.ptr(self.bits), self.bits.shape[1], _lib.ptr(self.ws1), self.ws1.numel(),
                                            B, HW, prec, fmt, s()), "ph_khead_onepass")
            # the in-call fallback: the two-pass kernels and the binarisation, PREDICATED on the one-pass launch's status word
            # (first word of ws1).  The persistent grid needs a workgroup resident on every CU; when another kernel holds CUs
            # beyond the hand-off bound the launch gives up, raises that word, and these launches -- which otherwise return at
            # once -- rewrite every output of the call.  No host round trip, valid under graph capture and replay.
            if _os.environ.get("PH_KHEAD_NO_FALLBACK"):       # timing experiments only: what the predicated launches cost
                return self._finish_run(lib, pk, s, B, HW, prec)
            _lib.check(lib.ph_khead_fused_if(_lib.ptr(self.f[0]), _lib.ptr(self.f[1]), _lib.ptr(self.f[2]), _lib.ptr(pk.wplanes),
                                             _lib.ptr(pk.gn), pk.groups, 1e-5, _lib.ptr(pk.init_frag), self.Nq,
                                             _lib.ptr(pk.seg_frag), _lib.ptr(pk.seg_bias), pk.n_seg, _lib.ptr(pk.dd_frag),
                                             _lib.ptr(pk.dd_bias), self.n_thing_cls, self.n_stuff, _lib.ptr(self.xp), _lib.ptr(self.dp),
                                             _lib.ptr(self.x_f32), _lib.ptr(self.dfe_f32), _lib.ptr(self.mask_preds),
                                             _lib.ptr(self.seg_preds), _lib.ptr(self.depth_pred), OUT_CODE[self.logit_dtype],
                                             _lib.ptr(self.ws1), _lib.ptr(self.ws), self.ws.numel(), B, HW, prec, fmt, s()),
                       "ph_khead_fused_if")
            _lib.check(lib.ph_binarize_if(_lib.ptr(self.mask_preds), OUT_CODE[self.logit_dtype], 0, _lib.ptr(self.bits), B, self.N, HW,
                                          _lib.ptr(self.ws1), s()), "ph_binarize_if")
        else:
            # conv1x1+GN+ReLU x3, x = sem + loc, and the static 1x1 convs on the normalised tiles (kernel_head.py:250-331):
            # init_kernels(loc) -> thing rows of mask_preds (:256), conv_seg(sem) -> seg_preds (:295) and its stuff rows ->
            # the remaining rows of mask_preds (:329-331), conv_direct_depth(dfe) -> depth_pred (:285)
            _lib.check(lib.ph_khead_fused(_lib.ptr(self.f[0]), _lib.ptr(self.f[1]), _lib.ptr(self.f[2]), _lib.ptr(pk.wplanes),
                                          _lib.ptr(pk.gn), pk.groups, 1e-5, _lib.ptr(pk.init_frag), self.Nq,
                                          _lib.ptr(pk.seg_frag), _lib.ptr(pk.seg_bias), pk.n_seg, _lib.ptr(pk.dd_frag),
                                          _lib.ptr(pk.dd_bias), self.n_thing_cls, self.n_stuff, _lib.ptr(self.xp), _lib.ptr(self.dp),
                                          _lib.ptr(self.x_f32), _lib.ptr(self.dfe_f32), _lib.ptr(self.mask_preds),
                                          _lib.ptr(self.seg_preds), _lib.ptr(self.depth_pred), _lib.ptr(self.ws), self.ws.numel(),
                                          B, HW, prec, fmt, s()), "ph_khead_fused")
            # object features: binarise the logits once (all rows: the decode stages start from these bits), pool x over the
            # THING rows (:314-320)
            _lib.check(lib.ph_binarize(_lib.ptr(self.mask_preds), 0, _lib.ptr(self.bits), B, self.N, HW, s()), "ph_binarize")
        self._finish_run(lib, pk, s, B, HW, prec)

    def _finish_run(self, lib, pk, s, B, HW, prec):
        # object features: pool x over the THING rows of the bit tensor (:314-320), add them to the kernels (:324-326)
        _lib.check(lib.ph_pool_rows(_lib.ptr(self.xp), None, _lib.ptr(self.bits), self.bits.shape[1], _lib.ptr(self.partial),
                                    B, self.Nq, HW, self.nsplit, prec, s()), "ph_pool_rows")
        _lib.check(lib.ph_khead_proposals(_lib.ptr(self.partial), self.nsplit, _lib.ptr(pk.w_init_f32),
                                          _lib.ptr(self.w_stuff),
                                          _lib.ptr(self.proposal), B, self.Nq, self.n_stuff, s()), "ph_khead_proposals")

    def timeouts(self):
        """one-pass form: number of workgroup time-outs since the plan was built (sticky across calls and graph replays;
        synchronises the stream).  A time-out costs the affected call its one-pass speed -- the predicated two-pass fallback
        inside `run` produces its results -- never the results."""
        if not self.onepass:
            return 0
        return int(_lib.load().ph_khead_onepass_timeouts(_lib.ptr(self.ws1), self.B, self.HW, _lib.stream_ptr()))

    def last_run_fell_back(self):
        """one-pass form: True if the most recent run gave up and was redone by the two-pass kernels (synchronises)"""
        return bool(self.onepass and _lib.load().ph_khead_onepass_status(_lib.ptr(self.ws1), self.B, _lib.stream_ptr()) != 0)

    def check_status(self):
        """kept for callers of round 3's API: returns `timeouts()`; nothing to raise any more, results are valid either way"""
        return self.timeouts()


# ---- SemanticFPNWrapper (N3) -------------------------------------------------------------------------------
def nhwc_ingest(x, add, prec, out):
    B, Cc, H, W = x.shape
    lib = _lib.load()
    _lib.check(lib.ph_nhwc_ingest(_lib.ptr(x), _lib.ptr(add), _lib.ptr(out), B, H * W, prec, _lib.stream_ptr()), "ph_nhwc_ingest")
    return out


def conv_nhwc(xp, pk, y, partial, B, H, W, prec):
    lib = _lib.load()
    _lib.check(lib.ph_conv_nhwc(_lib.ptr(xp), _lib.ptr(pk["wp"]), pk["wp"].shape[1], _lib.ptr(y), _lib.ptr(partial), pk["k"],
                                pk["s"], B, H, W, prec, _lib.stream_ptr()), "ph_conv_nhwc")


def gn_finalize(partial, stats, nwg, groups, HW, B, eps=1e-5):
    lib = _lib.load()
    _lib.check(lib.ph_gn_finalize(_lib.ptr(partial), _lib.ptr(stats), nwg, groups, HW, eps, B, _lib.stream_ptr()), "ph_gn_finalize")


def gn_apply(y, stats, pk, groups, mode, B, H, W, prec, planes=None, outf=None, accumulate=False):
    lib = _lib.load()
    _lib.check(lib.ph_gn_apply(_lib.ptr(y), _lib.ptr(stats), _lib.ptr(pk["gamma"]) if pk else None,
                               _lib.ptr(pk["beta"]) if pk else None, groups, mode, 1 if accumulate else 0, _lib.ptr(planes),
                               _lib.ptr(outf), B, H, W, prec, _lib.stream_ptr()), "ph_gn_apply")


def gn_sum_planes(ys, stats, pks, groups, planes, B, HW, prec):
    lib = _lib.load()
    n = len(ys)
    arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
    _lib.check(lib.ph_gn_sum_planes(arr(ys), arr(stats), arr([p["gamma"] for p in pks]), arr([p["beta"] for p in pks]), n, groups,
                                    _lib.ptr(planes), B, HW, prec, _lib.stream_ptr()), "ph_gn_sum_planes")


def gn_sum_cplanes(ys, stats, pks, groups, planes, B, HW, prec):
    """the level sum as channel planes [P,B,256,HWp] (ph_neck_out_convs' input)"""
    lib = _lib.load()
    n = len(ys)
    arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
    _lib.check(lib.ph_gn_sum_cplanes(arr(ys), arr(stats), arr([p["gamma"] for p in pks]), arr([p["beta"] for p in pks]), n, groups,
                                     _lib.ptr(planes), B, HW, prec, _lib.stream_ptr()), "ph_gn_sum_cplanes")


def neck_out_convs(planes, channels_last, wplanes, gn_affine, groups, out_planes, out_f32, ws, B, HW, prec, eps=1e-5):
    """conv_pred + 2 aux convs (1x1 conv + GN + ReLU each) of the level sum `planes` ([P,B,HW,256] when channels_last, else channel
    planes [P,B,256,HWp]); out_planes / out_f32: lists of 3 (None: not wanted)"""
    lib = _lib.load()
    op = out_planes if out_planes is not None else [None] * 3
    of = out_f32 if out_f32 is not None else [None] * 3
    _lib.check(lib.ph_neck_out_convs(_lib.ptr(planes), 1 if channels_last else 0, _lib.ptr(wplanes), _lib.ptr(gn_affine), groups, eps,
                                     _lib.ptr(op[0]), _lib.ptr(op[1]), _lib.ptr(op[2]), _lib.ptr(of[0]), _lib.ptr(of[1]), _lib.ptr(of[2]),
                                     _lib.ptr(ws), ws.numel() * ws.element_size(), B, HW, prec, _lib.stream_ptr()), "ph_neck_out_convs")


class NeckPlan:
    """buffers + launch sequence of SemanticFPNWrapper.forward for one (B, level shapes): channels-last bf16 planes
    between the convs, fp32 channels-last conv outputs (one per level for the fused level sum), three fp32 NCHW outputs"""

    def __init__(self, B, shapes, prec, device, tower_streams=True):
        self.B, self.shapes, self.prec = B, shapes, prec
        P = 2 if prec == _lib.PH_PREC_SPLIT else 1
        dev = torch.device(device)
        e = lambda shape, dt: torch.empty(shape, dtype=dt, device=dev)
        (h0, w0), (h1, w1) = shapes[0], shapes[1]
        self.Ho, self.Wo = h1, w1                                  # stride-8 output size
        if ((h0 + 1) // 2, (w0 + 1) // 2) != (h1, w1) or any(shapes[i + 1] != ((shapes[i][0] + 1) // 2, (shapes[i][1] + 1) // 2)
                                                             for i in range(1, 3)):
            raise _lib.PolyheadError(f"FPN level sizes {shapes} are not a stride-2 pyramid")
        big = max(h * w for h, w in shapes)
        self.xa = e((P, B, big, 256), torch.int16)                 # conv input planes (ping)
        self.xb = e((P, B, self.Ho * self.Wo, 256), torch.int16)   # conv input planes (pong, <= output size)
        self.ys = [e((B, self.Ho * self.Wo, 256), torch.float32) for _ in range(4)]   # last conv output of each level
        self.y = e((B, self.Ho * self.Wo, 256), torch.float32)     # other conv outputs (pre-norm)
        lib = _lib.load()
        self.partial = e((lib.ph_conv_nhwc_partial_floats(B, self.Ho, self.Wo),), torch.float32)
        self.lstats = [e((B, 256, 2), torch.float32) for _ in range(4)]
        self.stats = e((B, 256, 2), torch.float32)
        self.outs = [e((B, 256, self.Ho, self.Wo), torch.float32) for _ in range(3)]
        self.pouts = None                                          # plane outputs, allocated on first use
        # Round 3: the four level towers are independent until the level sum and the small ones leave most of the chip idle
        # (a 3x3 conv at 16 x 32 x 16 frames is 32 workgroups): each tower gets its own buffers and its own HIP stream, the
        # small launches run beside the stride-4 level's ingest + stride-2 conv.  PH_NECK_STREAMS=0 or the module attribute
        # `tower_streams = False`: one stream, shared buffers (needed when TWO pipelines are captured into one HIP graph: the
        # nested fork / join of 2 x 4 streams made hipStreamEndCapture segfault on ROCm 7.2).
        # (from 4 frames per call: one frame at a time -- the video loop -- is bound by the host's launch rate, where the stream
        # switches cost more than the overlap returns: cfg4 7.95 -> 9.08 ms per two frames with tower streams)
        _ns = _os.environ.get("PH_NECK_STREAMS", "1")
        self.multi = tower_streams and (B >= 4 or _ns == "2" or tower_streams == "always") and _ns != "0" and dev.type == "cuda"
        self.lv = None
        if self.multi:
            self.lv = []
            for lvl, (h, w) in enumerate(shapes):
                # the towers of levels 2 and 3 upsample up to the output size between their convs; levels 0 and 1 are a single conv
                n_small = self.Ho * self.Wo if lvl >= 2 else 1
                self.lv.append(dict(xa=e((P, B, max(h * w, n_small), 256), torch.int16), xb=e((P, B, n_small, 256), torch.int16),   # ping / pong
                                    y=e((B, n_small, 256), torch.float32), stats=e((B, 256, 2), torch.float32),
                                    partial=e((lib.ph_conv_nhwc_partial_floats(B, self.Ho, self.Wo),), torch.float32)))
        self._streams = None                                       # created on first use; not part of a copy of the plan
        self.skip_ingest = False                                   # set around a capture whose replays follow `ingest_frames`
        # Round 4: the three output convs (conv_pred + 2 aux convs) as stats / apply passes over the level sum in channel planes
        # (ph_neck_out_convs) instead of conv -> fp32 NHWC -> finalize -> apply per map.  PH_NECK_OUT2=0: the per-map form.
        # one-plane grades only: with hi / lo planes the recompute pass is three MFMAs per product and costs more than the fp32
        # round trip it removes (whole head 19.1 -> 20.4 ms per 16 frames at the parity grade, same box)
        self.out2 = _os.environ.get("PH_NECK_OUT2", "1") != "0" and (P == 1 or _os.environ.get("PH_NECK_OUT2") == "3")
        self.sc = self.ws2 = None
        self.out2_cplanes = _os.environ.get("PH_NECK_OUT2", "1") == "2"
        if self.out2:
            if self.out2_cplanes:
                self.sc = e((P, B, 256, hw_padded(self.Ho * self.Wo)), torch.int16)
            self.ws2 = e((lib.ph_neck_out_convs_workspace_bytes(B, self.Ho * self.Wo, 32) // 4 + 64,), torch.float32)

    def __getstate__(self):
        d = dict(self.__dict__)
        d["_streams"] = None               # HIP streams cannot be copied / pickled (copy.deepcopy of a module that holds a plan)
        return d

    def _conv_gn(self, xp, pk, H, W, groups, y, stats, partial=None, layout=0):
        """conv + statistics; returns the conv output size"""
        B, prec = self.B, self.prec | layout
        partial = self.partial if partial is None else partial
        k, s = pk["k"], pk["s"]
        Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
        conv_nhwc(xp, pk, y, partial, B, H, W, prec)
        nwg = _lib.load().ph_conv_nhwc_workgroups_b(k, s, Ho, Wo, prec, B)
        gn_finalize(partial, stats, nwg, groups, Ho * Wo, B)
        return Ho, Wo

    def _tower(self, lvl, feat, pk, groups, posenc, bufs):
        """one level: ingest -> (conv + GN + ReLU + x2 upsample)* -> last conv (its GroupNorm is applied by the level sum)"""
        B, prec = self.B, self.prec
        H, W = self.shapes[lvl]
        xa, xb, y, stats, partial = bufs["xa"], bufs["xb"], bufs["y"], bufs["stats"], bufs["partial"]
        convs = pk["levels"][lvl]
        # a level that starts with the stride-2 conv (level 0) hands it chunk-major planes (PH_PLANES_C16: the kernel's 16-channel
        # stages then read whole lines; PH_NECK_C16=0: channels-last, A/B timing)
        c16 = _lib.PH_PLANES_C16 if (convs[0]["s"] == 2 and convs[0]["k"] == 3 and prec != _lib.PH_PREC_SPLIT
                                     and _os.environ.get("PH_NECK_C16", "1") != "0") else 0
        if not self.skip_ingest:
            nhwc_ingest(feat, posenc, prec | c16, xa)
        src = xa
        for j, c in enumerate(convs):
            lay = c16 if j == 0 else 0
            if j + 1 < len(convs):      # every non-final conv of levels 2 and 3 is followed by an x2 upsample
                H, W = self._conv_gn(src, c, H, W, groups, y, stats, partial, lay)
                dst = xb if src is xa else xa
                gn_apply(y, stats, c, groups, _lib.PH_GN_UP2_PLANES, B, H, W, prec, planes=dst)
                H, W, src = 2 * H, 2 * W, dst
            else:
                H, W = self._conv_gn(src, c, H, W, groups, self.ys[lvl], self.lstats[lvl], partial, lay)
                if (H, W) != (self.Ho, self.Wo):
                    raise _lib.PolyheadError("level does not end at the stride-8 size")

    def can_ingest_frames(self):
        """True when the levels' conv input planes can be filled frame by frame AHEAD of `run` (`ingest_frames`): every level owns its
        buffers (the tower-stream form) and the grade has one plane"""
        return bool(self.multi) and self.prec != _lib.PH_PREC_SPLIT

    def ingest_frames(self, frames, pk, posenc, pos_level):
        """round 6 (video clips without the staging copy): the ingest step of `run` for `frames` = B one-frame level tuples (fp32
        [1, 256, h, w], contiguous), each written straight from the caller's tensor into its slice of the level's conv input planes -- on the
        CURRENT stream, outside any graph.  A `run(..)` with `skip_ingest` set (the captured graph) continues from those planes."""
        if not self.can_ingest_frames() or len(frames) != self.B:
            raise _lib.PolyheadError("NeckPlan.ingest_frames: needs the tower-stream form, a one-plane grade and B frames")
        lib, st = _lib.load(), _lib.stream_ptr()
        for lvl, (h, w) in enumerate(self.shapes):
            convs = pk["levels"][lvl]
            c16 = _lib.PH_PLANES_C16 if (convs[0]["s"] == 2 and convs[0]["k"] == 3 and _os.environ.get("PH_NECK_C16", "1") != "0") else 0
            xa = self.lv[lvl]["xa"]
            add = _lib.ptr(posenc) if lvl == pos_level and posenc is not None else None
            for b, f in enumerate(frames):
                t = f[lvl]
                if t.dtype != torch.float32 or not t.is_contiguous() or tuple(t.shape) != (1, 256, h, w) or t.device != xa.device:
                    raise _lib.PolyheadError("NeckPlan.ingest_frames: fp32 contiguous [1, 256, h, w] device tensors")
                _lib.check(lib.ph_nhwc_ingest(_lib.ptr(t), add, C.c_void_p(xa.data_ptr() + b * h * w * 256 * 2), 1, h * w, self.prec | c16, st),
                           "ph_nhwc_ingest(frame)")

    def run(self, feats, pk, groups, posenc, pos_level, to_planes=False):
        B, prec = self.B, self.prec
        if self.multi:
            if self._streams is None:
                self._streams = [torch.cuda.Stream(device=self.xa.device) for _ in range(4)]
            cur = torch.cuda.current_stream()
            for lvl in (0, 3, 2, 1):                # the longest chains first
                st = self._streams[lvl]
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    self._tower(lvl, feats[lvl], pk, groups, posenc if lvl == pos_level else None, self.lv[lvl])
            for st in self._streams:
                cur.wait_stream(st)
        else:
            shared = dict(xa=self.xa, xb=self.xb, y=self.y, stats=self.stats, partial=self.partial)
            for lvl in range(4):
                self._tower(lvl, feats[lvl], pk, groups, posenc if lvl == pos_level else None, shared)
        if to_planes and self.pouts is None:
            P = 2 if prec == _lib.PH_PREC_SPLIT else 1
            self.pouts = [torch.empty((P, B, 256, hw_padded(self.Ho * self.Wo)), dtype=torch.int16, device=self.xa.device)
                          for _ in range(3)]
        if self.out2 and len(pk["outs"]) == 3 and groups == 32:
            # level sum (channels-last planes, as before); statistics pass (three maps in one launch) + finalize + one apply launch
            # per map.  PH_NECK_OUT2=2: through channel planes (ph_gn_sum_cplanes; the transposing sum is 0.2 ms slower per 16 frames)
            if self.out2_cplanes:
                gn_sum_cplanes(self.ys, self.lstats, [pk["levels"][l][-1] for l in range(4)], groups, self.sc, B, self.Ho * self.Wo, prec)
            else:
                gn_sum_planes(self.ys, self.lstats, [pk["levels"][l][-1] for l in range(4)], groups, self.xb, B, self.Ho * self.Wo, prec)
            neck_out_convs(self.sc if self.out2_cplanes else self.xb, not self.out2_cplanes, pk["outs_w"], pk["outs_gn"], groups,
                           self.pouts if to_planes else None, None if to_planes else self.outs, self.ws2, B, self.Ho * self.Wo, prec)
            return self.pouts if to_planes else self.outs
        # sum over levels of ReLU(GN(.)) straight to conv input planes, then conv_pred / aux convs -> fp32 NCHW
        gn_sum_planes(self.ys, self.lstats, [pk["levels"][l][-1] for l in range(4)], groups, self.xb, B, self.Ho * self.Wo, prec)
        for i, c in enumerate(pk["outs"]):
            self._conv_gn(self.xb, c, self.Ho, self.Wo, groups, self.y, self.stats)
            if to_planes:       # bf16 channel planes, the decode path's feature format (KernelHead hand-off)
                gn_apply(self.y, self.stats, c, groups, _lib.PH_GN_TO_CPLANES, B, self.Ho, self.Wo, prec, planes=self.pouts[i])
            else:
                gn_apply(self.y, self.stats, c, groups, _lib.PH_GN_TO_NCHW, B, self.Ho, self.Wo, prec, outf=self.outs[i])
        return (self.pouts if to_planes else self.outs)[:len(pk["outs"])]


class DualDecodePlan:
    """`parts` part-batches (two by default) on as many HIP streams, each one phase behind the previous: the query
    kernels of a stage are a short, latency-bound chain on ~1 workgroup per CU, the pooling / conv / upsample kernels are
    HBM-bound; running one part's HBM phases underneath another's query phase (and vice versa) fills both.  Frames are
    independent, so the split changes nothing numerically.  Captured as ONE HIP graph with fork/join edges.  Measured on
    one box at 24 frames per part: 2 parts 13.1-13.3 k frames/s, 4 parts 13.4-13.8 k (the fill / drain of the skew is
    amortised over more parts), 6 or 8 parts no further gain."""

    def __init__(self, packs, B, N, H, W, prec, out_dtype=torch.float32, device="cuda:0", parts=2):
        assert B >= parts >= 2
        self.B, self.parts = B, parts
        base, rem = divmod(B, parts)
        self.sizes = [base + (1 if i < rem else 0) for i in range(parts)]
        self.halves = [DecodePlan(packs, n, N, H, W, prec, out_dtype, device) for n in self.sizes]
        for p in self.halves:
            p.shares_gpu = True          # query launches with the most rows per workgroup: the other parts' kernels run beside them
        self.graph = None

    def set_inputs(self, x, dfe, k0, q0, m0):
        o = 0
        before = [p.m0.dtype for p in self.halves]
        for p, n in zip(self.halves, self.sizes):
            p.set_inputs(x[o:o + n], dfe[o:o + n], k0[o:o + n], q0[o:o + n], m0[o:o + n])
            o += n
        if before != [p.m0.dtype for p in self.halves]:
            self.graph = None            # the parts re-allocated their mask-logit buffers: the captured graph is stale (capture again)

    def _issue(self, *streams):
        cur = torch.cuda.current_stream()
        prev = None
        for p, st in zip(self.halves, streams):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                if prev is not None:
                    st.wait_event(prev)      # a part starts once the previous one's ingest is done: a phase apart from then on
                p.ingest()
                prev = torch.cuda.Event()
                prev.record(st)
                p.stages()
        for st in streams:
            cur.wait_stream(st)

    def run(self):
        if not hasattr(self, "_streams"):
            self._streams = tuple(torch.cuda.Stream() for _ in range(self.parts))
        self._issue(*self._streams)

    def capture(self):
        self.run()
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._issue(*self._streams)
        return self.graph

    def replay(self):
        self.graph.replay()

    def outputs(self):
        o = [h.outputs() for h in self.halves]
        return {k: (None if o[0][k] is None else torch.cat([oi[k] for oi in o], 0)) for k in o[0]}
