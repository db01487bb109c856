"""SURVEY.md 8(f) row N3 -- `SemanticFPNWrapper` (polyphonic/funcs/semantic_fpn.py:16-235), the `localization_fpn`
of KernelHead (kernel_head.py:70,243), as shipped in configs/_base_/models/polyphonic_former.py:78-96:
levels 0..3, upsample_times 2, GN(32), sum fusion, SinePositionalEncoding on level 3, conv_pred + 2 aux convs.

Same registry name, constructor kwargs and state_dict keys as the reference (`convs_all_levels.{i}.conv{j}.conv.weight`,
`...gn.weight/bias`, `conv_pred.*`, `aux_convs.{i}.*`); all arithmetic runs in libpolyhead (csrc/ph_neck.hip):
channels-last implicit-GEMM convolutions on MFMA, GroupNorm statistics fused into the conv epilogue."""
import math

import torch
import torch.nn as nn

from . import _lib, engine as E
from .bricks import build_norm_layer
from .pack import pack_b32
from .registry import register_everywhere


class _ConvGN(nn.Module):
    """parameter container with ConvModule's tree (`conv`, `gn`): conv without bias (a norm follows), GN, ReLU"""

    def __init__(self, cin, cout, k, norm_cfg, stride=1):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2, bias=False)
        name, norm = build_norm_layer(norm_cfg, cout)
        assert name == "gn"
        self.gn = norm
        self.ksize, self.stride = k, stride


class _Level(nn.Module):
    """`convs_all_levels[i]`: named children conv0, conv1, ... (the reference adds parameter-free `upsample{j}`
    modules in between; they have no state)"""

    def __init__(self, convs):
        super().__init__()
        for j, c in enumerate(convs):
            self.add_module(f"conv{j}", c)
        self.n = len(convs)


def sine_positional_encoding(H, W, num_feats, temperature=10000, scale=2 * math.pi, eps=1e-6):
    """mmdet SinePositionalEncoding(normalize=True) for an empty ignore mask -> fp32 [2*num_feats, H, W]; a constant
    of the map size, computed once on the host (mmdet/models/utils/positional_encoding.py:56-91)"""
    y = torch.arange(1, H + 1, dtype=torch.float32).view(H, 1).expand(H, W)
    x = torch.arange(1, W + 1, dtype=torch.float32).view(1, W).expand(H, W)
    y = y / (y[-1:, :] + eps) * scale
    x = x / (x[:, -1:] + eps) * scale
    dim_t = torch.arange(num_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * (dim_t // 2) / num_feats)
    px, py = x[..., None] / dim_t, y[..., None] / dim_t
    px = torch.stack((px[..., 0::2].sin(), px[..., 1::2].cos()), dim=3).view(H, W, -1)
    py = torch.stack((py[..., 0::2].sin(), py[..., 1::2].cos()), dim=3).view(H, W, -1)
    return torch.cat((py, px), dim=2).permute(2, 0, 1).contiguous()


class SemanticFPNWrapper(nn.Module):
    # round 5: in training mode under autograd (`.train()`, grad mode on, a parameter or an input that requires grad) `forward` runs the differentiable
    # fp32 form (`train.neck_forward_train`: 3x3 conv / GroupNorm + ReLU / upsample nodes with hand-written backward), so
    # KernelHead.forward_train trains the neck -- and through its input gradients the FPN and the backbone -- as the reference does
    # (polyphonic_former.py:97-110).  Without autograd: the inference kernels on packed 16-bit weights, as before.
    differentiable = True

    def __init__(self, in_channels, feat_channels, out_channels, start_level, end_level, cat_coors=False,
                 positional_encoding=None, cat_coors_level=3, fuse_by_cat=False, return_list=False, upsample_times=3,
                 with_pred=True, num_aux_convs=0, act_cfg=dict(type="ReLU", inplace=True), out_act_cfg=dict(type="ReLU"),
                 conv_cfg=None, norm_cfg=None):
        super().__init__()
        ok = (in_channels == feat_channels == out_channels == 256 and start_level == 0 and end_level == 3 and
              upsample_times == 2 and not cat_coors and not fuse_by_cat and with_pred and norm_cfg is not None and
              norm_cfg.get("type") == "GN" and conv_cfg is None)
        if not ok:
            raise NotImplementedError("libpolyhead implements the shipped SemanticFPNWrapper configuration "
                                      "(256 channels, levels 0-3, upsample_times 2, GN, sum fusion)")
        if positional_encoding is not None and not (positional_encoding.get("type") == "SinePositionalEncoding" and
                                                    positional_encoding.get("normalize", False)):
            raise NotImplementedError("positional_encoding must be SinePositionalEncoding(normalize=True) or None")
        self.pos_cfg = dict(positional_encoding) if positional_encoding else None
        self.cat_coors_level, self.return_list, self.num_aux_convs = cat_coors_level, return_list, num_aux_convs
        self.groups = norm_cfg["num_groups"]
        C = 256
        mk = lambda k=3, s=1: _ConvGN(C, C, k, norm_cfg, stride=s)
        # semantic_fpn.py:75-150 for (start 0, end 3, upsample_times 2): conv counts 1 (stride 2), 1, 2, 3
        self.convs_all_levels = nn.ModuleList([_Level([mk(3, 2)]), _Level([mk()]), _Level([mk(), mk()]),
                                               _Level([mk(), mk(), mk()])])
        self.conv_pred = mk(1)
        self.aux_convs = nn.ModuleList([mk(1) for _ in range(num_aux_convs)])
        self.precision = "fp32"
        self._packs, self._plans, self._pos = {}, {}, {}

    def init_weights(self):
        for m in self.modules():                       # semantic_fpn.py:181-186
            if isinstance(m, nn.Conv2d):
                nn.init.normal_(m.weight, 0, 0.01)

    def set_precision(self, precision):
        assert precision in E.PREC
        self.precision = precision
        self._packs.clear(); self._plans.clear()

    # ---- packed parameters ---------------------------------------------------------------------------------
    def _pack(self, dev):
        # keyed on the parameter versions too: load_state_dict / init_weights / fine-tuning after a first forward re-pack
        key = (str(dev), self.precision, _lib.param_versions(self))
        if key not in self._packs:
            self._packs.clear()
            prec = E.KHEAD_PREC[self.precision]             # 'fp16': one fp16 plane of weights and activations (f16 MFMA)
            P = 2 if prec == _lib.PH_PREC_SPLIT else 1

            def one(m):
                w = m.conv.weight.detach().to("cpu", torch.float64)                   # [out][in][kh][kw]
                w2 = w.permute(0, 2, 3, 1).reshape(256, -1)                           # K order (tap, channel)
                planes = E._planes_of(w2, P, prec == _lib.PH_PREC_F16)                                          # [P][256][K]
                wp = torch.stack([pack_b32(planes[p]) for p in range(P)], 0).contiguous().to(dev)
                return dict(wp=wp, gamma=m.gn.weight.detach().float().contiguous().to(dev),
                            beta=m.gn.bias.detach().float().contiguous().to(dev), k=m.ksize, s=m.stride)
            lv = [[one(getattr(l, f"conv{j}")) for j in range(l.n)] for l in self.convs_all_levels]
            outs = [self.conv_pred] + list(self.aux_convs)
            pk = dict(levels=lv, outs=[one(m) for m in outs])
            if len(outs) == 3:      # the three output convs as [out][in] planes + GN affine: the operands of ph_neck_out_convs
                w3 = torch.stack([m.conv.weight.detach().to("cpu", torch.float64).reshape(256, 256) for m in outs], 0)
                pk["outs_w"] = E._planes_of(w3, P, prec == _lib.PH_PREC_F16).to(dev)                      # [P,3,256,256]
                pk["outs_gn"] = torch.stack([torch.stack([m.gn.weight.detach().float(), m.gn.bias.detach().float()], 0)
                                             for m in outs], 0).contiguous().to(dev)                       # [3,2,256]
            self._packs[key] = pk
        return self._packs[key]

    def _posenc(self, H, W, dev):
        key = (H, W, str(dev))
        if key not in self._pos:
            self._pos[key] = sine_positional_encoding(H, W, self.pos_cfg["num_feats"], self.pos_cfg.get("temperature", 10000),
                                                      self.pos_cfg.get("scale", 2 * math.pi), self.pos_cfg.get("eps", 1e-6)).to(dev)
        return self._pos[key]

    # ---- forward (semantic_fpn.py:198-235) ---------------------------------------------------------------------
    def forward_planes(self, inputs):
        """the same maps as `forward`, as the decode path's feature format: bf16 planes [P][B][256][HWp] (int16 tensors,
        hi or hi/lo) -- the internal hand-off to KernelHead (ph_khead_fused, PH_IN_PLANES).  Only with the aux convs
        (three outputs), which is the configuration KernelHead needs."""
        if self.num_aux_convs != 2:
            raise NotImplementedError("forward_planes needs the three outputs (num_aux_convs = 2)")
        return self.forward(inputs, _planes=True)

    def clip_plan(self, B, shapes, dev):
        """the device plan `forward` uses for B frames of these level sizes (None before the first such call)"""
        return self._plans.get((B, tuple(tuple(s) for s in shapes), str(dev), self.precision))

    def ingest_frames(self, frames, plan=None):
        """round 6: fills the plan's conv input planes from B one-frame level tuples, frame by frame and without a batched copy of the
        levels (video.VideoStreamRunner's borrowed clips; engine.NeckPlan.ingest_frames).  `plan`: the plan a captured graph replays
        (the caller holds it); default: this module's plan for that clip size, which must exist (one `forward` of B frames)."""
        t0 = frames[0][0]
        shapes = tuple(tuple(t.shape[-2:]) for t in frames[0][:4])
        if plan is None:
            plan = self.clip_plan(len(frames), shapes, t0.device)
        if plan is None:
            raise _lib.PolyheadError("SemanticFPNWrapper.ingest_frames: no plan for this clip size yet")
        add = self._posenc(*shapes[self.cat_coors_level], t0.device) if self.pos_cfg is not None else None
        plan.ingest_frames([f[:4] for f in frames], self._pack(t0.device), add, self.cat_coors_level)

    def forward(self, inputs, _planes=False):
        x0 = inputs[0]
        E._require_gpu(x0, "inputs[0]")
        if not _planes and self.training and torch.is_grad_enabled() and (any(p.requires_grad for p in self.parameters()) or
                                                        any(torch.is_tensor(t) and t.requires_grad for t in inputs[:4])):
            from . import train as T
            outs = T.neck_forward_train(self, inputs)
            if self.num_aux_convs > 0:
                return outs
            return [outs[0]] if self.return_list else outs[0]
        dev, B = x0.device, x0.shape[0]
        pk, prec, G = self._pack(dev), E.KHEAD_PREC[self.precision], self.groups
        shapes = tuple(tuple(t.shape[-2:]) for t in inputs[:4])
        plan = self._plans.get((B, shapes, str(dev), self.precision))
        if plan is None:
            # `_clip_towers` (set by video.VideoStreamRunner around the capture of a 2-3 frame clip launch): the four level towers on
            # their own streams also below 4 frames -- inside a HIP graph the forks cost the host nothing (engine.NeckPlan)
            ts = getattr(self, "tower_streams", True)
            if ts and B < 4 and getattr(self, "_clip_towers", False):
                ts = "always"
            plan = E.NeckPlan(B, shapes, prec, dev, tower_streams=ts)
            self._plans[(B, shapes, str(dev), self.precision)] = plan
        add = self._posenc(*shapes[self.cat_coors_level], dev) if self.pos_cfg is not None else None
        outs = plan.run([t.float().contiguous() for t in inputs[:4]], pk, G, add, self.cat_coors_level, to_planes=_planes)
        if _planes:
            return outs
        if self.num_aux_convs > 0:
            return outs
        return [outs[0]] if self.return_list else outs[0]


register_everywhere(SemanticFPNWrapper)
