"""The video frame loop (video.VideoStreamRunner, cfg3 geometry) for a kernel trace:
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/v -o v -- python tools/video_trace.py [frames] [eager|seq|eager+seq]
prints ms per frame; the per-kernel table (calls / frames = launches per frame) is the trace's kernel_stats.csv."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from polyphonicformer_amd import video as V

n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
graph = not (len(sys.argv) > 2 and "eager" in sys.argv[2])
pipelined = not (len(sys.argv) > 2 and "seq" in sys.argv[2])
bench.host_thread_policy()
dev = torch.device("cuda:0")
pipe, cfg, wl = bench._video_pipeline(dev, "fp16")
H8, W8 = wl["H"] * 8, wl["W"] * 8
g = torch.Generator().manual_seed(31)
base = [torch.randn(1, 256, H8 // s, W8 // s, generator=g).to(dev) for s in (4, 8, 16, 32)]
meta = [dict(img_shape=(H8, W8, 3), ori_shape=(H8, W8, 3), batch_input_shape=(H8, W8))]
runner = V.VideoStreamRunner(pipe, meta[0], graph=graph, pipelined=pipelined)
xs = [bench._video_frame(base, f, 6) for f in range(6)]
for f in range(6):
    runner.push(xs[f])
runner.flush()
pipe.assoc.init_tracker()
torch.cuda.synchronize()
t0 = time.perf_counter()
for f in range(n):
    runner.push(xs[f % 6])
runner.flush()
torch.cuda.synchronize()
print({"frames": n, "ms_per_frame": round((time.perf_counter() - t0) / n * 1e3, 3), "graph": graph, "pipelined": pipelined})
