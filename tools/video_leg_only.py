"""prints bench.video_leg (cfg3: module API, stream runner, batched stream runner) alone.  usage: python tools/video_leg_only.py [precision]"""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
bench.host_thread_policy() if hasattr(bench, "host_thread_policy") else None
out = bench.video_leg(dev, precision=sys.argv[1] if len(sys.argv) > 1 else "fp16")
print(json.dumps({k: v for k, v in out.items() if k.startswith("stream_runner") or k == "ms_per_frame"}, indent=1))
