"""GPU: cProfile of one training step (host side): where the wall time goes.  usage: python tools/train_hostprof.py"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from polyphonicformer_amd import train as T  # noqa: E402

wl = bench.WORKLOADS["cfg2"]
dev = torch.device("cuda:0")
orig = T.TrainStep.forward_backward
calls = []


def wrapped(self, *a, **k):
    calls.append((self, a, k))
    return orig(self, *a, **k)


T.TrainStep.forward_backward = wrapped
bench.train_leg(wl, dev, 1, B=2, steps=1)
T.TrainStep.forward_backward = orig
self, a, k = calls[0]
k = dict(k, backward=True)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    for p in self.parameters():
        p.grad = None
    orig(self, *a, **k)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
