"""times ph_query_stage (pre, post) for several frames-per-launch, precision modes (run on the GPU box)
usage: python tools/query_time.py [frames ...]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from polyphonicformer_amd import engine as E

wl = bench.WORKLOADS[os.environ.get("PH_QT_WORKLOAD", "cfg2")]
dev = torch.device("cuda:0")
N = wl["Nq"] + wl["n_stuff"]
frames = [int(a) for a in sys.argv[1:]] or [24, 48, 64, 96]
for prec in ("bf16", "fp16"):
    head = bench.build_head(wl, prec, torch.float16 if prec == "fp16" else torch.bfloat16, dev)
    for B in frames:
        plan = head._plan(B, N, wl["H"], wl["W"], dev)
        inp = bench.synth_inputs(wl, B, seed=1)
        g = [inp[k].to(dev) for k in ("x", "dfe", "k0", "q0", "m0")]
        g[0], g[1] = g[0].to(plan.mode.feat_dtype), g[1].to(plan.mode.feat_dtype)
        plan.set_inputs(*g)
        plan.run()
        torch.cuda.synchronize()
        t = {}
        for name, ph in (("pre", 1), ("post", 2), ("both", 3)):
            t[name] = bench.time_op(lambda ph=ph: E.query_stage(plan.partial, plan.bits, plan.k0, plan.q0, plan.packs[0], plan.N, plan.HW,
                                                                outs=plan.stage_out[0], workspace=plan.ws, phases=ph,
                                                                kern_fmt=plan.mode.kern_fmt, counts=plan.pcount), 20) * 1e3
        print(json.dumps({"prec": prec, "frames": B, "v1": os.environ.get("PH_QUERY_V1", "0"), "us": {k: round(v, 1) for k, v in t.items()},
                          "us_per_frame": round(t["both"] / B, 2)}), flush=True)
        del plan
        head._plans.clear()
        torch.cuda.empty_cache()
