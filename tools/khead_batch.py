import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
wl = bench.WORKLOADS["cfg2"]
for prec, odt in (("fp16", torch.float16), ("bf16", torch.bfloat16)):
    head = bench.build_head(wl, prec, odt, dev)
    for B in (16, 32, 48):
        r = bench.kernel_head_leg(wl, head, prec, odt, dev, B=B)
        print(prec, B, json.dumps({k: r[k] for k in r if k != "note"}), flush=True)
    del head
    torch.cuda.empty_cache()
