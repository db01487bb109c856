"""KernelHead post-neck (a1) alone at the cfg2 shape, 16 frames: eager and HIP-graph time per call; run under
`rocprofv3 --kernel-trace --stats` for the per-kernel split (tools/collect_profiles.sh)."""
import sys, torch, time
sys.path.insert(0, '.')
from polyphonicformer_amd.registry import HEADS
from polyphonicformer_amd import engine as E
import polyphonicformer_amd.kernel_head
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
wl = dict(Nq=100, n_thing=80, n_stuff=53, H=128, W=256)
L = 133
torch.manual_seed(1)
kh = HEADS.build(dict(type="KernelHead", num_proposals=100, num_classes=L, num_thing_classes=80,
                      num_stuff_classes=53, cat_stuff_mask=True, feat_downsample_stride=2, feat_refine=False,
                      use_binary=True, proposal_feats_with_obj=True, kernel_init_std=1, conv_normal_init=True,
                      loss_seg=dict(type="FocalLoss", use_sigmoid=True), localization_fpn=None))
kh.init_weights(); kh.eval().to(dev); kh.set_precision(sys.argv[2] if len(sys.argv) > 2 else 'bf16'); kh.emit_fp32_features = False
kplan = E.KernelHeadPlan(kh._get_pack(dev), B, 128, 256, 80, L, True, dev, want_f32=False)
g = torch.Generator().manual_seed(3)
kplan.set_inputs([torch.randn(B, 256, 128, 256, generator=g).relu().to(dev) for _ in range(3)])
kplan.run(); torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    kplan.run()
for f in (kplan.run, graph.replay):
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10): f()
    torch.cuda.synchronize()
    print('ms per call', (time.perf_counter() - t) * 100)
