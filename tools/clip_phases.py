"""GPU: where a clip's step goes in `VideoStreamRunner.records` (bench --workload cfg4): the heads of B frames in one launch (graph replay,
waited for), then per frame the merge (`DeviceMerge.finish`: accept loop + paste) and the record (boxes -> RoIAlign -> track head), each waited
for; and the un-instrumented `records` call.  usage: python tools/clip_phases.py [frames = 8] [precision = fp16]"""
import json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from polyphonicformer_amd import video as V
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
prec = sys.argv[2] if len(sys.argv) > 2 else "fp16"
bench.host_thread_policy()
dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
pipe, cfg, wl = bench._video_pipeline(dev, prec)
H8, W8 = wl["H"] * 8, wl["W"] * 8
g = torch.Generator().manual_seed(31)
base = [torch.randn(1, 256, H8 // s, W8 // s, generator=g).to(dev) for s in (4, 8, 16, 32)]
meta = dict(img_shape=(H8, W8, 3), ori_shape=(H8, W8, 3), batch_input_shape=(H8, W8))
runner = V.VideoStreamRunner(pipe, meta)
frames = [bench._video_frame(base, f, 6) for f in range(B)]
sync = torch.cuda.synchronize
for _ in range(3):
    runner.records(frames)
sync()
res = {"frames": B, "precision": prec, "clip_batch": runner.clip_batch(frames)}
t = []
for _ in range(5):
    sync(); t0 = time.perf_counter(); runner.records(frames); sync(); t.append(time.perf_counter() - t0)
res["records_ms_per_frame"] = round(sorted(t)[2] / B * 1e3, 3)
th, tm, tr = [], [], []
for _ in range(5):
    Bc = runner.clip_batch(frames)
    sync(); t0 = time.perf_counter()
    runner._start_heads(0, frames[:Bc]); sync()
    t1 = time.perf_counter()
    merged = []
    for b in range(Bc):
        merged.append(runner._merge(0, b)); sync()
    t2 = time.perf_counter()
    for b in range(Bc):
        pan_dev, info, _, _ = merged[b]
        pipe.assoc.record(runner._frame_levels(0, b), None, info, pan_dev); sync()
    t3 = time.perf_counter()
    th.append((t1 - t0) / Bc); tm.append((t2 - t1) / Bc); tr.append((t3 - t2) / Bc)
med = lambda v: round(sorted(v)[len(v) // 2] * 1e3, 3)
res.update(heads_ms_per_frame=med(th), merge_ms_per_frame=med(tm), record_ms_per_frame=med(tr))
print(json.dumps(res))
