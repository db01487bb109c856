"""GPU: the neck leg and the whole-head leg (bench.neck_leg / full_head_leg) in the fp16 and bf16 grades"""
import sys, json, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
wl = bench.WORKLOADS["cfg2"]
for prec, odt in (("fp16", torch.float16), ("bf16", torch.bfloat16), ("fp32", torch.float32)):
    try:
        r = bench.neck_leg(wl, prec, dev)
        print("neck", prec, json.dumps({k: r[k] for k in r if k != "note"}), flush=True)
        head = bench.build_head(wl, prec, odt, dev)
        r = bench.full_head_leg(wl, head, prec, dev)
        print("full", prec, json.dumps({k: r[k] for k in r if k != "note"}), flush=True)
        del head
    except Exception as e:
        import traceback; traceback.print_exc()
    torch.cuda.empty_cache()
