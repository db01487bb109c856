"""Parameter containers that reproduce the *module tree* (hence the state_dict key names) of the
mmcv 1.3.18 bricks the reference instantiates on the hot path, so published checkpoints load
unchanged.  They hold parameters only: arithmetic happens in libpolyhead."""
import math

import torch
import torch.nn as nn

from .registry import LOSSES


def build_norm_layer(cfg, num_features):
    cfg = dict(cfg)
    typ = cfg.pop("type")
    cfg.pop("requires_grad", None)
    cfg.setdefault("eps", 1e-5)
    if typ == "LN":
        return "ln", nn.LayerNorm(num_features, **cfg)
    if typ == "GN":
        return "gn", nn.GroupNorm(num_channels=num_features, **cfg)
    raise NotImplementedError(f"norm type {typ}")


class ConvModuleParams(nn.Module):
    """`conv` (+ `gn`) like mmcv.cnn.ConvModule: bias='auto' => conv bias iff no norm."""

    def __init__(self, in_channels, out_channels, kernel_size, norm_cfg=None, act_cfg=dict(type="ReLU"),
                 conv_cfg=None, stride=1, padding=0, **kw):
        super().__init__()
        if kernel_size != 1 or stride != 1 or padding != 0:
            raise NotImplementedError("libpolyhead implements the shipped 1x1 / stride-1 ConvModules only")
        self.conv = nn.Conv2d(in_channels, out_channels, 1, bias=norm_cfg is None)
        self.with_norm = norm_cfg is not None
        self.with_activation = act_cfg is not None
        if self.with_norm:
            name, norm = build_norm_layer(norm_cfg, out_channels)
            self.norm_name = name
            self.add_module(name, norm)


class FFNParams(nn.Module):
    """mmcv FFN tree: layers = Sequential(Sequential(Linear, act, Dropout), Linear, Dropout)"""

    def __init__(self, embed_dims, feedforward_channels, num_fcs=2, act_cfg=None, dropout=0.0, **kw):
        super().__init__()
        if num_fcs != 2 or dropout != 0.0:
            raise NotImplementedError("FFN with num_fcs=2 and dropout=0 only (the shipped config)")
        if act_cfg is not None and act_cfg.get("type", "ReLU") != "ReLU":
            raise NotImplementedError("FFN activation must be ReLU")
        self.layers = nn.Sequential(
            nn.Sequential(nn.Linear(embed_dims, feedforward_channels), nn.ReLU(inplace=True), nn.Dropout(0.0)),
            nn.Linear(feedforward_channels, embed_dims), nn.Dropout(0.0))


class MultiheadAttentionParams(nn.Module):
    """mmcv MultiheadAttention tree: `attn` = nn.MultiheadAttention(embed_dims, num_heads)"""

    def __init__(self, embed_dims, num_heads, attn_drop=0.0, **kw):
        super().__init__()
        if attn_drop != 0.0:
            raise NotImplementedError("attention dropout must be 0")
        self.attn = nn.MultiheadAttention(embed_dims, num_heads, 0.0)


class _LossStub(nn.Module):
    """Placeholder for the TRACK head's loss configs (its training is out of scope, track_head.py); the path's own losses
    are real classes (losses.py)."""

    def __init__(self, use_sigmoid=False, **kw):
        super().__init__()
        self.use_sigmoid = use_sigmoid
        self.cfg = dict(kw)

    def forward(self, *a, **k):
        raise NotImplementedError("training losses are outside the hot path this package implements")



def bias_init_with_prob(p):
    return float(-math.log((1 - p) / p))
