"""CPU restatement of the step before the hot path: `SemanticFPNWrapper.forward`
(polyphonic/funcs/semantic_fpn.py:198-235) as configured in configs/_base_/models/polyphonic_former.py:78-96, incl.
`SinePositionalEncoding` on level 3 (mmdet/models/utils/positional_encoding.py:56-91).

TEST INFRASTRUCTURE ONLY (tests/, oracle/gen_golden.py): the product path never imports this module.
Pinned by tests/golden/*neck*.npz, produced by the reference classes themselves (oracle/gen_golden_neck.py)."""
import math

import torch
import torch.nn.functional as F


def sine_positional_encoding(B, H, W, num_feats=128, temperature=10000, scale=2 * math.pi, eps=1e-6):
    """normalize=True, offset=0, empty ignore mask (semantic_fpn.py:202-208) -> [B, 2*num_feats, H, W]"""
    y = torch.arange(1, H + 1, dtype=torch.float32).view(1, H, 1).expand(B, H, W)      # cumsum of ones
    x = torch.arange(1, W + 1, dtype=torch.float32).view(1, 1, W).expand(B, H, W)
    y = y / (y[:, -1:, :] + eps) * scale
    x = x / (x[:, :, -1:] + eps) * scale
    dim_t = torch.arange(num_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * (dim_t // 2) / num_feats)
    px, py = x[..., None] / dim_t, y[..., None] / dim_t
    px = torch.stack((px[..., 0::2].sin(), px[..., 1::2].cos()), dim=4).view(B, H, W, -1)
    py = torch.stack((py[..., 0::2].sin(), py[..., 1::2].cos()), dim=4).view(B, H, W, -1)
    return torch.cat((py, px), dim=3).permute(0, 3, 1, 2)


def conv_gn_relu(x, sd, prefix, groups, stride=1, k=3):
    """mmcv ConvModule: conv (no bias: a norm follows) -> GroupNorm(eps 1e-5) -> ReLU"""
    x = F.conv2d(x, sd[prefix + "conv.weight"], None, stride=stride, padding=k // 2)
    x = F.group_norm(x, groups, sd[prefix + "gn.weight"], sd[prefix + "gn.bias"], eps=1e-5)
    return x.relu()


def up2(x):
    return F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)


def semantic_fpn(sd, feats, groups=32, prefix="", num_feats=128):
    """feats: the 4 FPN levels [B, C, H/4.., W/4..] (strides 4, 8, 16, 32) -> [out, aux0, aux1] at stride 8
    (start_level 0, end_level 3, upsample_times 2, num_aux_convs 2, fuse by sum)"""
    p = lambda s: prefix + s
    lv = []
    # level 0: ONE 3x3 stride-2 conv (semantic_fpn.py:76-107, end_level - upsample_times = 1 iteration)
    lv.append(conv_gn_relu(feats[0], sd, p("convs_all_levels.0.conv0."), groups, stride=2))
    # level 1: one conv, no upsample (:109-150 with i = 1)
    lv.append(conv_gn_relu(feats[1], sd, p("convs_all_levels.1.conv0."), groups))
    # level 2: conv, x2, conv
    x = conv_gn_relu(feats[2], sd, p("convs_all_levels.2.conv0."), groups)
    lv.append(conv_gn_relu(up2(x), sd, p("convs_all_levels.2.conv1."), groups))
    # level 3: + positional encoding, conv, x2, conv, x2, conv
    B, C, H, W = feats[3].shape
    x = feats[3] + sine_positional_encoding(B, H, W, num_feats)
    x = conv_gn_relu(x, sd, p("convs_all_levels.3.conv0."), groups)
    x = conv_gn_relu(up2(x), sd, p("convs_all_levels.3.conv1."), groups)
    lv.append(conv_gn_relu(up2(x), sd, p("convs_all_levels.3.conv2."), groups))
    s = lv[0] + lv[1] + lv[2] + lv[3]                                   # :221 fuse_by_cat=False
    outs = [conv_gn_relu(s, sd, p("conv_pred."), groups, k=1)]           # :223-224
    for i in range(2):                                                   # :229-232
        outs.append(conv_gn_relu(s, sd, p(f"aux_convs.{i}."), groups, k=1))
    return outs
