"""GPU: SemanticFPNWrapper.forward alone (bench.neck_leg) for a kernel trace.  usage: python tools/neck_only.py [precision] [B]"""
import sys, json, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
prec = sys.argv[1] if len(sys.argv) > 1 else "fp16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
print(json.dumps(bench.neck_leg(bench.WORKLOADS["cfg2"], prec, torch.device("cuda:0"), B=B)))
