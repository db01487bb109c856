"""GPU: the fused stage node against the CPU oracle under autograd for a few shapes; prints every gradient's error (debug aid)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as Hh  # noqa: E402
from oracle import poly_oracle as O  # noqa: E402
from polyphonicformer_amd import train as T  # noqa: E402
from polyphonicformer_amd.registry import HEADS  # noqa: E402
import polyphonicformer_amd.kernel_update_head, polyphonicformer_amd.kernel_updator  # noqa: F401,E402

gpu = torch.device("cuda:0")
cases = [(3, 153, 8, 16, 19, 8, 11, 2048), (3, 111, 8, 16, 19, 8, 11, 2048), (2, 153, 8, 16, 19, 8, 11, 2048), (2, 230, 8, 16, 19, 8, 11, 2048),
         (3, 153, 8, 16, 19, 8, 11, 256), (4, 100, 4, 8, 19, 8, 11, 2048), (3, 153, 4, 8, 19, 8, 11, 2048)]
if os.environ.get("QT_CASES"):
    cases = [tuple(int(v) for v in c.split(",")) for c in os.environ["QT_CASES"].split(";")]
for (B, N, H, W, L, nt, ns, F) in cases:
    h = HEADS.build(Hh.stage_cfg(256, F, 8, L, nt, ns))
    sd = Hh.seeded_fill({k: tuple(v.shape) for k, v in h.state_dict().items()}, B * 100 + N)
    h.load_state_dict(sd)
    h.to(gpu)
    g = torch.Generator().manual_seed(N)
    inp = dict(x=torch.randn(B, 256, H, W, generator=g), dfe=torch.randn(B, 256, H, W, generator=g), k=torch.randn(B, N, 256, generator=g),
               q=torch.randn(B, N, 256, generator=g), m=torch.randn(B, N, H, W, generator=g) - 0.3)
    cot = dict(cls=torch.randn(B, N, L, generator=g), mask=torch.randn(B, N, H, W, generator=g) * 0.1, obj=torch.randn(B, N, 256, generator=g),
               depth=torch.randn(B, N, H, W, generator=g) * 0.1, dobj=torch.randn(B, N, 256, generator=g))
    only = os.environ.get("QT_ONLY")
    if only:
        for k_ in cot:
            if k_ not in only.split(","):
                cot[k_] = cot[k_] * 0
    w = {k_: v.clone().requires_grad_(True) for k_, v in sd.items()}
    ci = {k_: (v.clone().requires_grad_(True) if k_ != "m" else v) for k_, v in inp.items()}
    r = O.update_stage(w, "", ci["x"], ci["k"], ci["m"], ci["q"], ci["dfe"])
    sum((r[n] * cot[n]).sum() for n in cot).backward()
    di = {k_: (v.to(gpu).requires_grad_(True) if k_ != "m" else v.to(gpu)) for k_, v in inp.items()}
    out = T.stage_forward(h, di["x"], di["dfe"], di["k"], di["m"], di["q"])
    names = ("cls", "mask", "obj", "depth", "dobj")
    sum((o * cot[n].to(gpu)).sum() for n, o in zip(names, out)).backward()
    torch.cuda.synchronize()
    print("CASE", (B, N, H, W, L, F), "fwd", {n: f"{Hh.rel_err(o.detach().cpu(), r[n].detach()):.1e}" for n, o in zip(names, out)})
    print("  inputs", {n: f"{Hh.rel_err(di[n].grad.cpu(), ci[n].grad):.1e}" for n in ("x", "dfe", "k", "q")})
    dk = (di["k"].grad.cpu() - ci["k"].grad).abs().amax(-1) / ci["k"].grad.abs().max()
    badrows = (dk.reshape(-1) > 1e-4).nonzero().flatten().tolist()
    print("  k-grad rows off by > 1e-4:", len(badrows), badrows[:40])
    pe = {n: Hh.rel_err(p.grad.cpu(), w[n].grad) for n, p in h.named_parameters()}
    bad = {n: f"{e:.1e}" for n, e in pe.items() if e > 1e-4}
    bad = dict(list(bad.items())[:6])
    print("  params over 1e-4 (first 6):", bad, "| worst ok", max([e for e in pe.values() if e <= 1e-4] or [0]))
