"""host-side phases of ONE module-API video frame (VideoFramePipeline.simple_test's graph path), a synchronisation after each --
the sum is above the un-instrumented per-frame time; it shows where the frame goes:  python tools/video_api_phases.py [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from polyphonicformer_amd import video as V

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
bench.host_thread_policy()
dev = torch.device("cuda:0")
pipe, cfg, wl = bench._video_pipeline(dev, "fp16")
H8, W8 = wl["H"] * 8, wl["W"] * 8
g = torch.Generator().manual_seed(31)
base = [torch.randn(1, 256, H8 // s, W8 // s, generator=g).to(dev) for s in (4, 8, 16, 32)]
meta = [dict(img_shape=(H8, W8, 3), ori_shape=(H8, W8, 3), batch_input_shape=(H8, W8))]
xs = [bench._video_frame(base, f, 6) for f in range(6)]
for f in range(6):
    pipe.simple_test(xs[f], meta)
r = next(iter(pipe._api_runners.values()))
sync = torch.cuda.synchronize
sync()
t0 = time.perf_counter()
for f in range(n):
    pipe.simple_test(xs[f % 6], meta)
sync()
plain = (time.perf_counter() - t0) / n * 1e3
acc = dict(heads=0.0, merge=0.0, assoc=0.0, download=0.0)
for f in range(n):
    x = xs[f % 6]
    sync(); t = time.perf_counter()
    r._check_weights()
    r._start_heads(0, [x])
    sync(); t1 = time.perf_counter()
    pan_dev, info, _, d_final = r._merge(0)
    sync(); t2 = time.perf_counter()
    sem, trk = pipe.assoc.step_device(r._frame_levels(0), pan_dev, info)
    sync(); t3 = time.perf_counter()
    host = [torch.empty(t_.shape, dtype=t_.dtype, pin_memory=True) for t_ in (sem, trk, d_final)]
    for h, t_ in zip(host, (sem, trk, d_final)):
        h.copy_(t_, non_blocking=True)
    sync(); t4 = time.perf_counter()
    acc["heads"] += t1 - t; acc["merge"] += t2 - t1; acc["assoc"] += t3 - t2; acc["download"] += t4 - t3
print({"ms_per_frame_plain": round(plain, 3), **{k: round(v / n * 1e3, 3) for k, v in acc.items()},
       "map_dtypes": [str(t_.dtype) for t_ in (sem, trk, d_final)], "map_MB": round(sum(t_.numel() * t_.element_size() for t_ in (sem, trk, d_final)) / 1e6, 1),
       "segments": len(info)})
