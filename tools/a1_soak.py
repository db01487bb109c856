"""soak: the one-pass KernelHead launch beside decode work on a second stream, status checked -- no hand-off may time out"""
import sys, time, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")        # run from the repository root
import bench
from polyphonicformer_amd import engine as E
from polyphonicformer_amd.registry import HEADS
import polyphonicformer_amd.kernel_head
dev = torch.device("cuda:0")
wl = bench.WORKLOADS["cfg2"]
L = wl["n_thing"] + wl["n_stuff"]
torch.manual_seed(1)
kh = HEADS.build(dict(type="KernelHead", num_proposals=wl["Nq"], num_classes=L, num_thing_classes=wl["n_thing"],
                      num_stuff_classes=wl["n_stuff"], cat_stuff_mask=True, feat_downsample_stride=2, feat_refine=False,
                      use_binary=True, proposal_feats_with_obj=True, kernel_init_std=1, conv_normal_init=True,
                      loss_seg=dict(type="FocalLoss", use_sigmoid=True)))
kh.init_weights(); kh.eval().to(dev); kh.set_precision("fp16")
H, W, B = wl["H"], wl["W"], 8
g = torch.Generator().manual_seed(3)
feats = [torch.randn(B, 256, H, W, generator=g).relu().to(dev) for _ in range(3)]
plan = E.KernelHeadPlan(kh._get_pack(dev), B, H, W, wl["n_thing"], L, True, dev, want_f32=False)
plan.set_inputs(feats)
head = bench.build_head(wl, "fp16", torch.float16, dev)
N = wl["Nq"] + wl["n_stuff"]
dp = head._plan(B, N, H, W, dev)
inp = bench.synth_inputs(wl, B, seed=1)
dp.set_inputs(inp["x"].to(dev).half(), inp["dfe"].to(dev).half(), inp["k0"].to(dev), inp["q0"].to(dev), inp["m0"].to(dev))
side = torch.cuda.Stream()
plan.run(); torch.cuda.synchronize()
ref = {k: getattr(plan, k).clone() for k in ("mask_preds", "bits", "xp", "proposal")}
t0 = time.time(); n = 0
while time.time() - t0 < 40:
    with torch.cuda.stream(side):
        dp.run()                      # HBM-bound and query kernels of the decode on another stream, concurrently
    for _ in range(20):
        plan.run(); n += 1
    assert plan.timeouts() == 0
    for k, v in ref.items():
        assert torch.equal(getattr(plan, k), v), k
torch.cuda.synchronize()
print("soak:", n, "one-pass launches beside concurrent decode launches, no time-out, results bit-identical")
