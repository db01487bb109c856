"""GPU: the a1 + a6 leg (bench.kernel_head_leg) in the three grades.  usage: python tools/khead_leg.py"""
import sys, json, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
wl = bench.WORKLOADS["cfg2"]
for prec, odt in (("fp16", torch.float16), ("bf16", torch.bfloat16), ("fp32", torch.float32)):
    head = bench.build_head(wl, prec, odt, dev)
    try:
        r = bench.kernel_head_leg(wl, head, prec, odt, dev)
        print(prec, json.dumps({k: r[k] for k in r if k != "note"}), flush=True)
    except Exception as e:
        print(prec, "error", repr(e))
    del head
    torch.cuda.empty_cache()
