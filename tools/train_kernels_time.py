"""GPU: the map-sized products of the training step (csrc/ph_train.hip) at cfg2's training sizes: time and HBM rate.
usage: python tools/train_kernels_time.py [B]"""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from polyphonicformer_amd import train as T
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
H, W, N, C = 128, 256, 153, 256
g = torch.Generator().manual_seed(0)
x = torch.randn(B, C, H, W, generator=g).to(dev)
k = torch.randn(B, N, C, generator=g).to(dev)
gm = torch.randn(B, N, H, W, generator=g).to(dev)
HW = H * W
rows = []
t = bench.time_op(lambda: T.rows_x_map(k, x), 10) * 1e3
rows.append(("rows_x_map  conv forward   [N x C] . [C x HW]", t, B * (C + N) * HW * 4))
t = bench.time_op(lambda: T.rows_x_map(k.transpose(1, 2), gm), 10) * 1e3
rows.append(("rows_x_map  d/dX of conv   [C x N] . [N x HW]", t, B * (C + N) * HW * 4))
t = bench.time_op(lambda: T.rows_x_map(k.transpose(1, 2), gm, binarize_x=True), 10) * 1e3
rows.append(("rows_x_map  d/dX of pool   [C x N] . bin[N x HW]", t, B * (C + N) * HW * 4))
t = bench.time_op(lambda: T.map_x_mapT(gm, x), 10) * 1e3
rows.append(("map_x_map_t d/dkernels     [N x HW] . [HW x C]", t, B * (C + N) * HW * 4))
t = bench.time_op(lambda: T.map_x_mapT(gm, x, binarize_g=True), 10) * 1e3
rows.append(("map_x_map_t pool forward   bin[N x HW] . [HW x C]", t, B * (C + N) * HW * 4))
up = torch.randn(B, N, 2 * H, 2 * W, generator=g).to(dev)
src = torch.randn(B, N, H, W, generator=g).to(dev).requires_grad_(True)
with torch.enable_grad():
    y = T.upsample2x(src)
t = bench.time_op(lambda: torch.autograd.grad(y, src, up, retain_graph=True), 10) * 1e3
rows.append(("upsample2x_bwd [N x 2H x 2W] -> [N x H x W]", t, B * N * HW * 4 * 5))
for name, us, nbytes in rows:
    print(json.dumps({"kernel": name, "B": B, "us": round(us, 1), "GBps": round(nbytes / us / 1e3, 1)}))
