import sys, time, torch, os
sys.path.insert(0,'/root/repo')
import bench
from oracle import poly_oracle as O
wl = bench.WORKLOADS["cfg2"]
from polyphonicformer_amd.registry import HEADS
torch.manual_seed(0)
import polyphonicformer_amd.kernel_update, polyphonicformer_amd.kernel_update_head, polyphonicformer_amd.kernel_updator
L = wl["n_thing"]+wl["n_stuff"]
head = HEADS.build(dict(type="KernelUpdateIterHead", num_stages=3, assign_stages=3, stage_loss_weights=[1]*3, num_proposals=100, num_thing_classes=80, num_stuff_classes=53, mask_head=bench.stage_cfg(L,80,53,2048)))
head.init_weights()
sd = {k: v.detach() for k, v in head.state_dict().items()}
inp = bench.synth_inputs(wl, 1, 1)
print("cpu_count", os.cpu_count())
os.system("lscpu | grep -E 'Model name|Socket|Core|Thread' ")
for nt in (8, 16, 32, 64, 128):
    torch.set_num_threads(nt)
    with torch.no_grad():
        O.iter_head_mask_preds(sd, 3, inp["x"], inp["k0"], inp["m0"], inp["q0"], inp["dfe"])
        t=time.time(); O.iter_head_mask_preds(sd, 3, inp["x"], inp["k0"], inp["m0"], inp["q0"], inp["dfe"]); dt=time.time()-t
    print(nt, "threads:", round(dt,3), "s/frame", flush=True)
