"""times get_panoptic (a7) at the cfg2 size on one frame (tuning helper)"""
import sys, time, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import bench, helpers as Hh
from polyphonicformer_amd import panoptic as Pn
from polyphonicformer_amd.registry import ConfigDict
dev = torch.device("cuda:0")
wl = bench.WORKLOADS["cfg2"]
N, L = 153, 133
g = torch.Generator().manual_seed(0)
cls = torch.rand(N, L, generator=g).to(dev)
m_up = torch.randn(N, 256, 512, generator=g).to(dev).to(torch.bfloat16)
d_up = torch.randn(N, 256, 512, generator=g).to(dev).to(torch.bfloat16)
d0 = torch.randn(1, 256, 512, generator=g).to(dev)
class H:
    merge_joint, num_proposals, num_thing_classes = True, 100, 80
    test_cfg = ConfigDict(max_per_img=100, merge_stuff_thing=dict(overlap_thr=0.6, instance_score_thr=0.3))
    mask_head = [type("S", (), {"depth_act_mode": "sigmoid"})()]
meta = Hh.img_meta(1024, 2048)
for it in range(3):
    torch.cuda.synchronize(); t = time.time()
    out = Pn.get_panoptic(H, cls, m_up, d_up, d0, meta)
    torch.cuda.synchronize(); print("get_panoptic 1024x2048, K=153: %.2f ms, %d segments" % ((time.time() - t) * 1e3, len(out[2][1])))
