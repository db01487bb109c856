"""Track embedding head and RoI extraction -- drop-ins for `QuasiDenseMaskEmbedHeadGTMask`
(polyphonic/video/track_heads.py:12-102, inference side) and the `SingleRoIExtractor` + RoIAlign call of
`PolyphonicVideo._track_forward` (polyphonic_former_video.py:408-419), on libpolyhead (csrc/ph_track.hip).
Same registry name, constructor kwargs and state_dict keys (convs.{i}.conv.weight, convs.{i}.gn.*, fcs.0.*,
fc_embed.*)."""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib, engine as E
from .bricks import _LossStub
from .pack import pack_b_fragments
from .registry import LOSSES, register_everywhere

for _n in ("MultiPosCrossEntropyLoss", "L2Loss"):
    if _n not in LOSSES:
        LOSSES.register_module(name=_n, module=type(_n, (_LossStub,), {}))


def segment_boxes(pan, nseg):
    """int32 id map [H,W] on the GPU, ids 1..nseg -> (rois [nseg,5], extent boxes [nseg,4]) fp32 on the GPU"""
    E._require_gpu(pan, "panoptic map")
    lib = _lib.load()
    pan = pan.to(torch.int32).contiguous()
    H, W = pan.shape
    rois = torch.empty((nseg, 5), dtype=torch.float32, device=pan.device)
    ext = torch.empty((nseg, 4), dtype=torch.float32, device=pan.device)
    ws = torch.empty((lib.ph_segment_boxes_workspace_bytes(nseg),), dtype=torch.uint8, device=pan.device)
    _lib.check(lib.ph_segment_boxes(_lib.ptr(pan), H, W, nseg, _lib.ptr(rois), _lib.ptr(ext), _lib.ptr(ws), ws.numel(),
                                    _lib.stream_ptr()), "ph_segment_boxes")
    return rois, ext


def roi_extract(feats, rois, prec, strides=(4, 8, 16, 32), finest_scale=56.0, want_f32=False):
    """feats: list of fp32 [1,256,H_l,W_l] GPU tensors (the FPN levels), rois [n,5] GPU.
    Returns channels-last bf16 planes int16 [P,n,49,256] (and fp32 [n,256,7,7] if asked)."""
    lib = _lib.load()
    n, dev = rois.shape[0], rois.device
    feats = [f.float().contiguous() for f in feats]
    for f in feats:
        if f.shape[0] != 1 or f.shape[1] != 256:
            raise _lib.PolyheadError("roi_extract: one image, 256 channels per level")
    P = 2 if prec == _lib.PH_PREC_SPLIT else 1
    out = torch.empty((P, n, 49, 256), dtype=torch.int16, device=dev)
    f32 = torch.empty((n, 256, 7, 7), dtype=torch.float32, device=dev) if want_f32 else None
    L = len(feats)
    ptrs = (C.c_void_p * L)(*[f.data_ptr() for f in feats])
    hw = (C.c_int32 * (2 * L))(*[v for f in feats for v in f.shape[-2:]])
    sc = (C.c_float * L)(*[1.0 / s for s in strides[:L]])
    rois_f = rois.float().contiguous()                 # named: alive until the launch is queued
    _lib.check(lib.ph_roi_align_fpn(ptrs, hw, sc, L, _lib.ptr(rois_f), n, finest_scale, _lib.ptr(out),
                                    _lib.ptr(f32), prec, _lib.stream_ptr()), "ph_roi_align_fpn")
    return (out, f32) if want_f32 else out


def _planes(w64, P):
    w = w64.to(torch.float32)
    hi = w.to(torch.bfloat16)
    out = [hi.view(torch.int16)]
    if P == 2:
        out.append((w - hi.float()).to(torch.bfloat16).view(torch.int16))
    return torch.stack(out, 0).contiguous()


class QuasiDenseMaskEmbedHeadGTMask(nn.Module):

    def __init__(self, num_convs=4, num_fcs=1, roi_feat_size=7, in_channels=256, conv_out_channels=256,
                 fc_out_channels=1024, embed_channels=256, conv_cfg=None, norm_cfg=None, softmax_temp=-1,
                 loss_track=None, loss_track_aux=None):
        super().__init__()
        if not (roi_feat_size == 7 and in_channels == 256 and conv_out_channels == 256 and num_fcs == 1 and num_convs >= 1
                and norm_cfg is not None and norm_cfg.get("type") == "GN" and fc_out_channels % 16 == 0
                and embed_channels % 16 == 0):
            raise NotImplementedError("libpolyhead implements the shipped track head (4 x conv3x3+GN+ReLU on 7x7x256, one fc)")
        self.num_convs, self.num_fcs, self.roi_feat_size = num_convs, num_fcs, roi_feat_size
        self.in_channels, self.conv_out_channels = in_channels, conv_out_channels
        self.fc_out_channels, self.embed_channels = fc_out_channels, embed_channels
        self.norm_cfg, self.softmax_temp = norm_cfg, softmax_temp
        self.groups = norm_cfg.get("num_groups", 32)
        self.convs = nn.ModuleList()
        for _ in range(num_convs):
            m = nn.Module()
            m.conv = nn.Conv2d(256, 256, 3, padding=1, bias=False)
            m.gn = nn.GroupNorm(self.groups, 256)
            self.convs.append(m)
        self.fcs = nn.ModuleList([nn.Linear(256 * 49, fc_out_channels)])
        self.fc_embed = nn.Linear(fc_out_channels, embed_channels)
        self.precision = "fp32"
        self._pack = None

    def init_weights(self):
        for m in self.fcs:
            nn.init.xavier_uniform_(m.weight)
            nn.init.constant_(m.bias, 0)
        nn.init.normal_(self.fc_embed.weight, 0, 0.01)
        nn.init.constant_(self.fc_embed.bias, 0)

    def _get_pack(self, device):
        prec = E.PREC[self.precision]
        ver = _lib.param_versions(self)
        key = (prec, str(device), ver)
        if self._pack is None or self._pack[0] != key:
            P = 2 if prec == _lib.PH_PREC_SPLIT else 1
            d = lambda t: t.detach().to("cpu", torch.float64)
            convs = []
            for m in self.convs:
                w = d(m.conv.weight).permute(0, 2, 3, 1).reshape(256, 9 * 256)        # K order (tap, channel)
                convs.append(_planes(pack_b_fragments(w), P).to(device))
            # fc consumes the NCHW flatten (ci*49 + pos) of the reference; our activations are (pos*256 + ci)
            wfc = d(self.fcs[0].weight).reshape(self.fc_out_channels, 256, 49).permute(0, 2, 1).reshape(self.fc_out_channels, -1)
            pk = dict(convs=convs, fc=_planes(pack_b_fragments(wfc), P).to(device),
                      emb=_planes(pack_b_fragments(d(self.fc_embed.weight)), P).to(device),
                      gn=[(m.gn.weight.detach().float().contiguous().to(device), m.gn.bias.detach().float().contiguous().to(device))
                          for m in self.convs],
                      fc_b=self.fcs[0].bias.detach().float().contiguous().to(device),
                      emb_b=self.fc_embed.bias.detach().float().contiguous().to(device), prec=prec, P=P)
            self._pack = (key, pk)
        return self._pack[1]

    def forward_planes(self, x_cl):
        """x_cl: channels-last bf16 planes int16 [P,n,49,256] (what `roi_extract` produces) -> embeddings [n,256] fp32"""
        lib, dev = _lib.load(), x_cl.device
        pk = self._get_pack(dev)
        P, n = x_cl.shape[0], x_cl.shape[1]
        if P != pk["P"]:
            raise _lib.PolyheadError("RoI feature planes were produced in a different precision than the head's")
        prec, s = pk["prec"], _lib.stream_ptr
        M, F_ = n * 49, self.fc_out_channels
        # split-K GEMMs (a dozen RoIs leave 16-68 workgroups otherwise); the convs gather their 3x3 patches in the operand loads
        ws = torch.empty((max(lib.ph_gemm_rows_workspace_bytes(M, 256, 2304), lib.ph_gemm_rows_workspace_bytes(n, F_, 49 * 256),
                              lib.ph_gemm_rows_workspace_bytes(n, self.embed_channels, F_)),), dtype=torch.uint8, device=dev)
        y = torch.empty((M, 256), dtype=torch.float32, device=dev)
        cur = x_cl.contiguous()
        for wp, (ga, be) in zip(pk["convs"], pk["gn"]):
            _lib.check(lib.ph_gemm_rows_splitk(_lib.ptr(cur), 1, _lib.ptr(wp), wp.shape[1], None, 0, _lib.ptr(y), None, M, 256, 2304, prec,
                                               _lib.ptr(ws), ws.numel(), s()), "ph_gemm_rows_splitk(conv)")
            nxt = torch.empty((P, n, 49, 256), dtype=torch.int16, device=dev)
            _lib.check(lib.ph_gn_relu_cl(_lib.ptr(y), _lib.ptr(ga), _lib.ptr(be), self.groups, 1e-5, _lib.ptr(nxt), n, prec, s()),
                       "ph_gn_relu_cl")
            cur = nxt
        h = torch.empty((P, n, F_), dtype=torch.int16, device=dev)
        _lib.check(lib.ph_gemm_rows_splitk(_lib.ptr(cur), 0, _lib.ptr(pk["fc"]), pk["fc"].shape[1], _lib.ptr(pk["fc_b"]), 1, None, _lib.ptr(h),
                                           n, F_, 49 * 256, prec, _lib.ptr(ws), ws.numel(), s()), "ph_gemm_rows_splitk(fc)")
        out = torch.empty((n, self.embed_channels), dtype=torch.float32, device=dev)
        _lib.check(lib.ph_gemm_rows_splitk(_lib.ptr(h), 0, _lib.ptr(pk["emb"]), pk["emb"].shape[1], _lib.ptr(pk["emb_b"]), 0, _lib.ptr(out),
                                           None, n, self.embed_channels, F_, prec, _lib.ptr(ws), ws.numel(), s()),
                   "ph_gemm_rows_splitk(fc_embed)")
        return out

    def forward(self, x):
        """track_heads.py:92-102: x fp32 [n,256,7,7] (RoI features) -> [n, embed_channels]"""
        E._require_gpu(x, "roi feats")
        n = x.shape[0]
        P = 2 if E.PREC[self.precision] == _lib.PH_PREC_SPLIT else 1
        xc = x.float().permute(0, 2, 3, 1).reshape(n, 49, 256)            # layout change only (channels last)
        hi = xc.to(torch.bfloat16)
        planes = [hi.view(torch.int16)]
        if P == 2:
            planes.append((xc - hi.float()).to(torch.bfloat16).view(torch.int16))
        return self.forward_planes(torch.stack(planes, 0).contiguous())

    def loss(self, *a, **k):
        raise NotImplementedError("training is outside the implemented path")


register_everywhere(QuasiDenseMaskEmbedHeadGTMask)
