"""GPU: wall time of the neck's training forward + backward at cfg2's size (B images 1024 x 2048: FPN levels 256x512 ... 32x64).
usage: python tools/neck_train_time.py [B]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import helpers as Hh  # noqa: E402
from polyphonicformer_amd.registry import NECKS  # noqa: E402
import polyphonicformer_amd.semantic_fpn  # noqa: F401,E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda:0")
neck = NECKS.build(dict(type="SemanticFPNWrapper", in_channels=256, feat_channels=256, out_channels=256, start_level=0, end_level=3,
                        upsample_times=2, positional_encoding=dict(type="SinePositionalEncoding", num_feats=128, normalize=True),
                        cat_coors=False, cat_coors_level=3, fuse_by_cat=False, return_list=False, num_aux_convs=2,
                        norm_cfg=dict(type="GN", num_groups=32, requires_grad=True)))
neck.init_weights()
neck.to(dev).train()
feats = [f.to(dev).requires_grad_(True) for f in Hh.fpn_inputs(seed=1, B=B, C=256, H0=256, W0=512)]


def step(backward=True):
    for p in neck.parameters():
        p.grad = None
    outs = neck(feats)
    if backward:
        sum(o.sum() for o in outs).backward()


for _ in range(2):
    step()
torch.cuda.synchronize()
res = {}
for name, bw in (("forward_backward_ms", True), ("forward_ms", False)):
    t0 = time.perf_counter()
    for _ in range(5):
        step(bw)
    torch.cuda.synchronize()
    res[name] = round((time.perf_counter() - t0) / 5 * 1e3, 2)
# 7 3x3 convs: level-0 stride-2 + 4 at 128x256 + 2 at 64x128 + 1 at 32x64, x (forward + input gradient + weight gradient)
flops = B * 2 * 256 * 256 * 9 * (4 * 128 * 256 + 2 * 64 * 128 + 32 * 64 + 128 * 256) * 3
res.update(images=B, conv3x3_TFLOPs_algorithmic=round(flops / (res["forward_backward_ms"] * 1e-3) / 1e12, 1),
           note="SemanticFPNWrapper training forward + backward (all parameters and the four FPN inputs), fp32 NCHW, 3 x bf16-split MFMA products")
print(json.dumps(res))
