"""Host orchestration of the decode hot path on one GPU: owns the device buffers (torch tensors are
used for allocation and streams only) and issues the libpolyhead kernels on the current stream.

Per frame-batch the launch sequence is

    ingest(x), ingest(depth_feats), binarize(mask_preds)
    for s in 0..S-1:   pool -> query_stage(pre, post) -> dynconv (bits for s < S-1, logits for s = S-1)
    dynconv(depth) ; upsample2x(mask) ; upsample2x(depth)

No step synchronises or allocates once a `DecodePlan` exists, so the whole sequence can be captured
into a HIP graph (`DecodePlan.capture`)."""
import ctypes as C

import torch

from . import _lib
from .pack import pack_stage

PREC = {"bf16": _lib.PH_PREC_BF16, "split": _lib.PH_PREC_SPLIT, "fp32": _lib.PH_PREC_SPLIT}


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


def default_nsplit(B, HW):
    nchunks = hw_padded(HW) // 128
    ns = max(1, -(-768 // (4 * B)))
    return int(min(ns, 32, max(1, nchunks)))


# ---- thin op wrappers (each = one C-ABI call) ---------------------------------------------------
def ingest(x, prec, out=None):
    """fp32 [B,256,H,W] -> bf16 planes int16 [P,B,256,HWp]"""
    _require_gpu(x, "x")
    B, Cc, H, W = x.shape
    if Cc != 256:
        raise _lib.PolyheadError("libpolyhead supports 256 channels (the shipped configs)")
    x = x.contiguous().float()
    P = 2 if prec == _lib.PH_PREC_SPLIT else 1
    if out is None:
        out = torch.empty((P, B, 256, hw_padded(H * W)), dtype=torch.int16, device=x.device)
    lib = _lib.load()
    _lib.check(lib.ph_ingest_features(_lib.ptr(x), _lib.ptr(out), B, H * W, prec, _lib.stream_ptr()), "ph_ingest_features")
    return out


def binarize(m, out=None):
    """fp32 mask logits [B,N,H,W] -> mask bits int32 [B,Npad,HWp/32]"""
    _require_gpu(m, "mask_preds")
    B, N, H, W = m.shape
    m = m.contiguous().float()
    if out is None:
        out = torch.empty((B, n_padded(N), hw_padded(H * W) // 32), dtype=torch.int32, device=m.device)
    lib = _lib.load()
    _lib.check(lib.ph_binarize(_lib.ptr(m), _lib.ptr(out), B, N, H * W, _lib.stream_ptr()), "ph_binarize")
    return out


def pool(xp, dp, bits, N, HW, prec, nsplit=None, out=None):
    B = xp.shape[1]
    if nsplit is None:
        nsplit = default_nsplit(B, HW)
    if out is None:
        out = torch.empty((B, nsplit, n_padded(N), 512), dtype=torch.float32, device=xp.device)
    lib = _lib.load()
    _lib.check(lib.ph_pool(_lib.ptr(xp), _lib.ptr(dp), _lib.ptr(bits), _lib.ptr(out), B, N, HW, nsplit, prec,
                           _lib.stream_ptr()), "ph_pool")
    return out


def query_stage(partial, bits, k_in, q_in, pack, N, HW, cls_sigmoid=False, outs=None, workspace=None, phases=3):
    B, nsplit = partial.shape[0], partial.shape[1]
    dev = partial.device
    prec = pack.prec
    P = 2 if prec == _lib.PH_PREC_SPLIT else 1
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
    _lib.check(lib.ph_query_stage(_lib.ptr(partial), nsplit, _lib.ptr(bits), _lib.ptr(k_in), _lib.ptr(q_in),
                                  _lib.ptr(pack.wb), _lib.ptr(pack.wf), C.byref(pack.lay),
                                  _lib.ptr(outs["obj"]), _lib.ptr(outs["dobj"]), _lib.ptr(outs["cls"]),
                                  1 if cls_sigmoid else 0, _lib.ptr(outs["kern"]), _lib.ptr(outs["kbias"]),
                                  _lib.ptr(workspace), workspace.numel(), B, N, HW, prec, phases, _lib.stream_ptr()),
               "ph_query_stage")
    return outs


def dynconv(planes, kern, kbias, branch, N, HW, prec, bits_out=None, logits_out=None, out_dtype=_lib.PH_OUT_F32):
    """kern [P,2,B,Npad,256], kbias [2,B,Npad]; `branch` selects mask (0) or depth (1)."""
    B, Npad = kern.shape[2], kern.shape[3]
    lib = _lib.load()
    kptr = C.c_void_p(kern.data_ptr() + branch * B * Npad * 256 * 2)
    bptr = C.c_void_p(kbias.data_ptr() + branch * B * Npad * 4)
    _lib.check(lib.ph_dynconv(_lib.ptr(planes), kptr, 2 * B * Npad * 256, bptr, _lib.ptr(bits_out),
                              _lib.ptr(logits_out), out_dtype, B, N, HW, prec, _lib.stream_ptr()), "ph_dynconv")
    return bits_out if bits_out is not None else logits_out


def upsample2x(src, out=None):
    """[..., H, W] fp32 or bf16 -> [..., 2H, 2W] (bilinear, align_corners=False)"""
    _require_gpu(src, "src")
    src = src.contiguous()
    H, W = src.shape[-2:]
    planes = src.numel() // (H * W)
    if out is None:
        out = torch.empty(tuple(src.shape[:-2]) + (2 * H, 2 * W), dtype=src.dtype, device=src.device)
    dt = _lib.PH_OUT_F32 if src.dtype == torch.float32 else _lib.PH_OUT_BF16
    if src.dtype not in (torch.float32, torch.bfloat16):
        raise _lib.PolyheadError("upsample2x: fp32 or bf16 only")
    lib = _lib.load()
    _lib.check(lib.ph_upsample2x(_lib.ptr(src), _lib.ptr(out), dt, planes, H, W, _lib.stream_ptr()), "ph_upsample2x")
    return out


# ---- the S-stage plan ------------------------------------------------------------------------------
class DecodePlan:
    """All buffers for `simple_test_mask_preds` at one (B, N, H, W, precision, output dtype)."""

    def __init__(self, packs, B, N, H, W, prec, out_dtype=torch.float32, device="cuda:0", nsplit=None):
        self.packs, self.S = packs, len(packs)
        self.B, self.N, self.H, self.W, self.HW = B, N, H, W, H * W
        self.prec, self.out_dtype = prec, out_dtype
        self.nsplit = nsplit or default_nsplit(B, self.HW)
        dev = torch.device(device)
        P = 2 if prec == _lib.PH_PREC_SPLIT else 1
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
        self.xp = e((P, B, 256, HWp), torch.int16)
        self.dp = e((P, B, 256, HWp), torch.int16)
        self.bits = e((B, Npad, HWp // 32), torch.int32)
        self.partial = e((B, self.nsplit, Npad, 512), torch.float32)
        self.ws = e((_lib.load().ph_query_workspace_bytes(B, N, prec),), torch.uint8)
        self.stage_out = [dict(obj=e((B, N, 256), torch.float32), dobj=e((B, N, 256), torch.float32),
                               cls=e((B, N, L), torch.float32), kern=e((P, 2, B, Npad, 256), torch.int16),
                               kbias=e((2, B, Npad), torch.float32)) for _ in range(self.S)]
        # outputs
        self.mask = e((B, N, H, W), out_dtype)
        self.depth = e((B, N, H, W), out_dtype)
        self.mask_up = e((B, N, 2 * H, 2 * W), out_dtype)
        self.depth_up = e((B, N, 2 * H, 2 * W), out_dtype)
        self.graph = None

    @property
    def out_code(self):
        return _lib.PH_OUT_F32 if self.out_dtype == torch.float32 else _lib.PH_OUT_BF16

    def set_inputs(self, x, dfe, k0, q0, m0):
        self.x.copy_(x)
        self.dfe.copy_(dfe)
        self.k0.copy_(k0.reshape(self.B, self.N, 256))
        self.q0.copy_(q0.reshape(self.B, self.N, 256))   # materialises the stride-0 expand view
        self.m0.copy_(m0)

    def ingest(self):
        ingest(self.x, self.prec, out=self.xp)
        ingest(self.dfe, self.prec, out=self.dp)
        binarize(self.m0, out=self.bits)

    def stages(self):
        k, q = self.k0, self.q0
        for s in range(self.S):
            last = s == self.S - 1
            pool(self.xp, self.dp, self.bits, self.N, self.HW, self.prec, self.nsplit, out=self.partial)
            o = query_stage(self.partial, self.bits, k, q, self.packs[s], self.N, self.HW, cls_sigmoid=last,
                            outs=self.stage_out[s], workspace=self.ws)
            if not last:
                dynconv(self.xp, o["kern"], o["kbias"], 0, self.N, self.HW, self.prec, bits_out=self.bits)
            else:
                dynconv(self.xp, o["kern"], o["kbias"], 0, self.N, self.HW, self.prec, logits_out=self.mask,
                        out_dtype=self.out_code)
                dynconv(self.dp, o["kern"], o["kbias"], 1, self.N, self.HW, self.prec, logits_out=self.depth,
                        out_dtype=self.out_code)
            k, q = o["obj"], o["dobj"]
        upsample2x(self.mask, out=self.mask_up)
        upsample2x(self.depth, out=self.depth_up)

    def run(self):
        """one pass: ingest + S stages + final upsample, on the current stream"""
        self.ingest()
        self.stages()

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
        return dict(obj=last["obj"], dobj=last["dobj"], cls=last["cls"], mask=self.mask, depth=self.depth,
                    mask_up=self.mask_up, depth_up=self.depth_up)
