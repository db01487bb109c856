"""GPU: a1 (KernelHead post-neck, one-pass form) alone at cfg2, HIP-graph replay, ms per call of `frames` frames; with
PH_KHEAD_NO_FALLBACK=1 in the environment the predicated two-pass launches behind the one-pass kernel are left out (timing only).
usage: python tools/a1_time.py [frames=16] [grade=fp16] [logits=fp32|fp16] [inputs=f32|planes]"""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polyphonicformer_amd.registry import HEADS
from polyphonicformer_amd import engine as E
import polyphonicformer_amd.kernel_head  # noqa: F401
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
grade = sys.argv[2] if len(sys.argv) > 2 else "fp16"
ldt = torch.float16 if (len(sys.argv) > 3 and sys.argv[3] == "fp16") else torch.float32
L = 133
torch.manual_seed(1)
kh = HEADS.build(dict(type="KernelHead", num_proposals=100, num_classes=L, num_thing_classes=80, num_stuff_classes=53,
                      cat_stuff_mask=True, feat_downsample_stride=2, feat_refine=False, use_binary=True, proposal_feats_with_obj=True,
                      kernel_init_std=1, conv_normal_init=True, loss_seg=dict(type="FocalLoss", use_sigmoid=True), localization_fpn=None))
kh.init_weights(); kh.eval().to(dev); kh.set_precision(grade); kh.emit_fp32_features = False
kplan = E.KernelHeadPlan(kh._get_pack(dev), B, 128, 256, 80, L, True, dev, want_f32=False, logit_dtype=ldt)
g = torch.Generator().manual_seed(3)
planes = len(sys.argv) > 4 and sys.argv[4] == "planes"      # the neck's hand-off: one 16-bit plane [1][B][256][HWp] per map
feats = [torch.randn(B, 256, 128, 256, generator=g).relu().to(dev) for _ in range(3)]
if planes:
    fdt = torch.float16 if grade == "fp16" else torch.bfloat16
    feats = [f.to(fdt).view(torch.int16).reshape(1, B, 256, 128 * 256).contiguous() for f in feats]
    assert E.hw_padded(128 * 256) == 128 * 256
kplan.set_inputs(feats)
kplan.run(); torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    kplan.run()
for _ in range(5):
    graph.replay()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
s.record()
for _ in range(30):
    graph.replay()
e.record()
torch.cuda.synchronize()
print(json.dumps({"a1_ms_per_call": round(s.elapsed_time(e) / 30, 4), "frames": B, "grade": grade, "logits": str(ldt), "inputs": "planes" if planes else "fp32 NCHW", "onepass": kplan.onepass,
                  "fallback_launches": not os.environ.get("PH_KHEAD_NO_FALLBACK"), "timeouts": kplan.timeouts()}))
