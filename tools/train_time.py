"""GPU: wall time of one training step (bench.train_leg) and its kernel mix.  usage: python tools/train_time.py [workload] [B]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
print(json.dumps(bench.train_leg(wl, torch.device("cuda:0"), 1, B=B)))
