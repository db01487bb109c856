"""GPU: the video leg alone (bench.video_leg).  usage: python tools/video_leg_only.py"""
import sys, json, torch
sys.path.insert(0, ".")
import bench
print(json.dumps(bench.video_leg(torch.device("cuda:0"))))
