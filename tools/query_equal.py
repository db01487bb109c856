"""runs ph_query_stage (pre + post) with the library given as argv[1] (or the package's) and prints a checksum of every output: two
libraries whose query kernels keep the same arithmetic order print identical lines.  usage: python tools/query_equal.py [lib] """
import os, sys, hashlib, torch
sys.path.insert(0, ".")
from polyphonicformer_amd import _lib
if len(sys.argv) > 1 and sys.argv[1] != "-":
    _lib.LIB_PATH = sys.argv[1]
import bench
from polyphonicformer_amd import engine as E
dev = torch.device("cuda:0")
for wlname, B, prec in (("cfg3", 1, "fp16"), ("cfg3", 3, "fp16"), ("cfg2", 24, "bf16"), ("cfg2", 5, "fp16"), ("cfg5", 7, "fp16")):
    wl = bench.WORKLOADS[wlname]
    N = wl["Nq"] + wl["n_stuff"]
    head = bench.build_head(wl, prec, torch.float16 if prec == "fp16" else torch.bfloat16, dev)
    plan = head._plan(B, N, wl["H"], wl["W"], dev)
    inp = bench.synth_inputs(wl, B, seed=1)
    g = [inp[k].to(dev) for k in ("x", "dfe", "k0", "q0", "m0")]
    if (wl["H"] * wl["W"]) % 128 == 0:
        g[0], g[1] = g[0].to(plan.mode.feat_dtype), g[1].to(plan.mode.feat_dtype)
    plan.set_inputs(*g)
    plan.run()
    torch.cuda.synchronize()
    o = plan.outputs()
    h = {k: hashlib.md5(v.detach().float().cpu().numpy().tobytes()).hexdigest()[:10] for k, v in o.items() if v is not None}
    t = {}
    for name, ph in (("pre", 1), ("post", 2)):
        t[name] = round(bench.time_op(lambda ph=ph: E.query_stage(plan.partial, plan.bits, plan.k0, plan.q0, plan.packs[0], plan.N, plan.HW,
                        outs=plan.stage_out[0], workspace=plan.ws, phases=ph, kern_fmt=plan.mode.kern_fmt, counts=plan.pcount), 20) * 1e3, 1)
    print(wlname, B, prec, h, t, flush=True)
    del plan, head
