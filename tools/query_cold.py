"""ph_query_stage (pre / post) at few frames per launch with the L2 cold (a 1 GB fill in front of every launch, as between two
stages of the decode) against back to back (weights of the previous launch still cached): python tools/query_cold.py [frames ...]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from polyphonicformer_amd import engine as E

wl = bench.WORKLOADS[os.environ.get("PH_QT_WORKLOAD", "cfg3")]
dev = torch.device("cuda:0")
N = wl["Nq"] + wl["n_stuff"]
frames = [int(a) for a in sys.argv[1:]] or [1, 3]
flush = torch.empty(256 << 20, dtype=torch.float32, device=dev)
for prec in ("fp16",):
    head = bench.build_head(wl, prec, torch.float16, dev)
    for B in frames:
        plan = head._plan(B, N, wl["H"], wl["W"], dev)
        inp = bench.synth_inputs(wl, B, seed=1)
        g = [inp[k].to(dev) for k in ("x", "dfe", "k0", "q0", "m0")]
        g[0], g[1] = g[0].to(plan.mode.feat_dtype), g[1].to(plan.mode.feat_dtype)
        plan.set_inputs(*g)
        plan.run()
        torch.cuda.synchronize()
        out = {}
        for name, ph in (("pre", 1), ("post", 2)):
            fn = lambda: E.query_stage(plan.partial, plan.bits, plan.k0, plan.q0, plan.packs[0], plan.N, plan.HW, outs=plan.stage_out[0],
                                       workspace=plan.ws, phases=ph, kern_fmt=plan.mode.kern_fmt, counts=plan.pcount)
            for cold in (False, True):
                ts = []
                for _ in range(12):
                    if cold:
                        flush.fill_(1.0)
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record(); fn(); e.record()
                    torch.cuda.synchronize()
                    ts.append(s.elapsed_time(e) * 1e3)
                out[name + ("_cold" if cold else "_warm")] = round(sorted(ts[2:])[len(ts[2:]) // 2], 1)
        print(json.dumps({"prec": prec, "workload": os.environ.get("PH_QT_WORKLOAD", "cfg3"), "frames": B, "us": out}), flush=True)
        del plan
        head._plans.clear()
        torch.cuda.empty_cache()
