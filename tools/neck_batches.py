"""SemanticFPNWrapper at cfg2's FPN sizes for a range of batch sizes, both output forms (fp32 NCHW / channel planes), checked against
the per-map output stage (PH_NECK_OUT2=0 semantics via plan.out2 = False).  usage: python tools/neck_batches.py [precision] [B ...]"""
import sys, torch
sys.path.insert(0, ".")
from polyphonicformer_amd.registry import NECKS
import polyphonicformer_amd.semantic_fpn  # noqa: F401
dev = torch.device("cuda:0")
prec = sys.argv[1] if len(sys.argv) > 1 else "fp16"
Bs = [int(b) for b in sys.argv[2:]] or [1, 2, 3, 5, 7]
torch.manual_seed(2)
m = NECKS.build(dict(type="SemanticFPNWrapper", in_channels=256, feat_channels=256, out_channels=256, start_level=0, end_level=3,
                     upsample_times=2, positional_encoding=dict(type="SinePositionalEncoding", num_feats=128, normalize=True),
                     cat_coors=False, cat_coors_level=3, fuse_by_cat=False, return_list=False, num_aux_convs=2,
                     norm_cfg=dict(type="GN", num_groups=32, requires_grad=True)))
m.init_weights(); m.eval().to(dev); m.set_precision(prec)
g = torch.Generator().manual_seed(4)
for B in Bs:
    feats = [torch.randn(B, 256, 256 >> i, 512 >> i, generator=g).to(dev) for i in range(4)]
    for planes in (False, True):
        new = [o.clone() for o in (m.forward_planes(feats) if planes else m(feats))]
        torch.cuda.synchronize()
        for p in m._plans.values():
            p.out2 = False
        old = [o.clone() for o in (m.forward_planes(feats) if planes else m(feats))]
        torch.cuda.synchronize()
        for p in m._plans.values():
            p.out2 = True
        if planes:
            f = lambda t: t[0].view(torch.float16 if prec == "fp16" else torch.bfloat16).float()
            err = max(float((f(a) - f(b)).abs().max() / f(b).abs().max()) for a, b in zip(new, old))
        else:
            err = max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(new, old))
        print(f"B={B} planes={planes}: max rel diff new vs per-map form {err:.2e}", flush=True)
