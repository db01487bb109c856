"""GPU: the video leg alone (bench.video_leg) in the fp16 and bf16 grades.  usage: python tools/video_leg_only.py"""
import sys, json, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
for prec in ("fp16", "bf16", "fp32"):
    r = bench.video_leg(torch.device("cuda:0"), precision=prec)
    print(prec, json.dumps({k: r[k] for k in r if k != "note"}), flush=True)
