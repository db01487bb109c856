"""GPU: one KernelUpdateHead stage in TRAINING form as a single autograd node (`train._Stage`: hard-mask pooling, the fused
query side of csrc/ph_qtrain.hip, the dynamic convolutions) against the CPU oracle's `update_stage` under torch autograd --
every output, and the gradient of every parameter and input, for random cotangents.  The oracle applies feat_transform as
written (kernel_update_head.py:224-226); the device path folds it, so its weight / bias gradients check the folding too."""
import pytest
import torch

import helpers as Hh
from oracle import poly_oracle as O
from polyphonicformer_amd import train as T
from polyphonicformer_amd.registry import HEADS
import polyphonicformer_amd.kernel_update_head  # noqa: F401
import polyphonicformer_amd.kernel_updator  # noqa: F401

pytestmark = pytest.mark.gpu


def _head(gpu, L, n_thing, n_stuff, F, seed):
    h = HEADS.build(Hh.stage_cfg(256, F, 8, L, n_thing, n_stuff))
    shapes = {k: tuple(v.shape) for k, v in h.state_dict().items()}
    sd = Hh.seeded_fill(shapes, seed)
    return h.to(gpu), sd


# the six ReLUs of a stage in the order `O.update_stage` evaluates them, and the bias that shifts each one's input
RELU_BIAS = ["kernel_update_conv.fc_norm.bias", "kernel_update_conv_depth.fc_norm.bias", "ffn.layers.0.0.bias", "ffn_depth.layers.0.0.bias",
             "cls_fcs.1.bias", "mask_fcs.1.bias"]


def condition_relus(sd, inp, margin=5e-5, rounds=20):
    """A ReLU input within rounding of 0 is a HARD decision: the device (other summation order, ~1e-6) may take the other
    side, and one flipped FFN unit moves a whole row of d W_1 (a unit that is active on a handful of rows) -- with 2 x 459 x 2048
    FFN activations a run has two or three such elements.  The comparison is about arithmetic, so the fixture is conditioned:
    biases are nudged (+2e-3 on the offending channel) until no ReLU input of the oracle's forward lies within `margin` of 0."""
    import torch.nn.functional as Fn
    real = Fn.relu
    for _ in range(rounds):
        seen = []
        Fn.relu = lambda t, *a, **k: (seen.append(t.detach()), real(t, *a, **k))[1]
        try:
            with torch.no_grad():
                O.update_stage(sd, "", inp["x"], inp["k"], inp["m"], inp["q"], inp["dfe"])
        finally:
            Fn.relu = real
        assert len(seen) == len(RELU_BIAS)
        dirty = False
        for t, name in zip(seen, RELU_BIAS):
            close = (t.abs() < margin).reshape(-1, t.shape[-1]).any(0)
            if close.any():
                sd[name] = sd[name] + 2e-3 * close.float()
                dirty = True
        if not dirty:
            return sd
    raise AssertionError("could not condition the fixture")


@pytest.mark.parametrize("B,N,H,W,L,nt,ns,F", [(2, 111, 6, 10, 19, 8, 11, 2048), (1, 37, 5, 7, 19, 8, 11, 256), (3, 153, 8, 16, 133, 80, 53, 2048), (3, 153, 8, 16, 19, 8, 11, 2048)])
def test_stage_node_vs_oracle_autograd(gpu, B, N, H, W, L, nt, ns, F):
    head, sd = _head(gpu, L, nt, ns, F, seed=B * 100 + N)
    g = torch.Generator().manual_seed(N)
    inp = dict(x=torch.randn(B, 256, H, W, generator=g), dfe=torch.randn(B, 256, H, W, generator=g),
               k=torch.randn(B, N, 256, generator=g), q=torch.randn(B, N, 256, generator=g),
               m=torch.randn(B, N, H, W, generator=g) - 0.3)
    cot = dict(cls=torch.randn(B, N, L, generator=g), mask=torch.randn(B, N, H, W, generator=g) * 0.1, obj=torch.randn(B, N, 256, generator=g),
               depth=torch.randn(B, N, H, W, generator=g) * 0.1, dobj=torch.randn(B, N, 256, generator=g))
    sd = condition_relus(sd, inp)
    head.load_state_dict(sd)
    # ---- oracle + torch autograd on the CPU
    with torch.enable_grad():
        w = {k_: v.clone().requires_grad_(True) for k_, v in sd.items()}
        ci = {k_: (v.clone().requires_grad_(True) if k_ != "m" else v) for k_, v in inp.items()}
        r = O.update_stage(w, "", ci["x"], ci["k"], ci["m"], ci["q"], ci["dfe"])
        sum((r[n] * cot[n]).sum() for n in cot).backward()
    # ---- the device node
    with torch.enable_grad():
        di = {k_: (v.to(gpu).requires_grad_(True) if k_ != "m" else v.to(gpu)) for k_, v in inp.items()}
        for p in head.parameters():
            p.grad = None
        out = T.stage_forward(head, di["x"], di["dfe"], di["k"], di["m"], di["q"])
        names = ("cls", "mask", "obj", "depth", "dobj")
        sum((o * cot[n].to(gpu)).sum() for n, o in zip(names, out)).backward()
    torch.cuda.synchronize()
    err = {n: Hh.rel_err(o.detach().cpu(), r[n].detach()) for n, o in zip(names, out)}
    print("stage node forward vs oracle:", {k_: f"{v:.1e}" for k_, v in err.items()})
    assert max(err.values()) < 2e-5, err
    gerr = {n: Hh.rel_err(di[n].grad.cpu(), ci[n].grad) for n in ("x", "dfe", "k", "q")}
    print("input gradients:", {k_: f"{v:.1e}" for k_, v in gerr.items()})
    assert max(gerr.values()) < 1e-4, gerr
    perr = {}
    for name, p in head.named_parameters():
        assert p.grad is not None, name
        perr[name] = Hh.rel_err(p.grad.cpu(), w[name].grad)
    worst = sorted(perr.items(), key=lambda kv: -kv[1])[:5]
    print("parameter gradients, worst:", [(k_, f"{v:.1e}") for k_, v in worst])
    assert worst[0][1] < 1e-4, worst


def test_stage_node_is_deterministic(gpu):
    """fixed summation orders: two runs give identical bits (outputs and gradients)"""
    head, sd = _head(gpu, 19, 8, 11, 2048, seed=5)
    head.load_state_dict(sd)
    g = torch.Generator().manual_seed(3)
    B, N, H, W = 2, 111, 6, 10
    x, dfe = torch.randn(B, 256, H, W, generator=g).to(gpu), torch.randn(B, 256, H, W, generator=g).to(gpu)
    k, q, m = torch.randn(B, N, 256, generator=g).to(gpu), torch.randn(B, N, 256, generator=g).to(gpu), torch.randn(B, N, H, W, generator=g).to(gpu)
    runs = []
    for _ in range(2):
        with torch.enable_grad():
            for p in head.parameters():
                p.grad = None
            kk = k.clone().requires_grad_(True)
            out = T.stage_forward(head, x, dfe, kk, m, q)
            sum(o.sum() for o in out).backward()
        runs.append([o.detach().clone() for o in out] + [kk.grad.clone()] + [p.grad.clone() for p in head.parameters()])
    for a, b in zip(*runs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("B,H,W,L,nt", [(2, 8, 16, 19, 8), (3, 7, 11, 19, 8), (1, 16, 24, 133, 80)])
def test_rpn_node_vs_oracle_autograd(gpu, B, H, W, L, nt):
    """KernelHead after the neck in training form (`train._Rpn`: towers with GroupNorm + ReLU, static convs, x = sem + loc,
    hard-mask pooling) against the oracle's `kernel_head_post_neck` under torch autograd: the six outputs and the gradient of
    every parameter and of the three input maps for random cotangents (ragged 7 x 11: no 16-byte aligned rows)"""
    import polyphonicformer_amd.kernel_head  # noqa: F401
    kh = HEADS.build(dict(type="KernelHead", num_proposals=100, num_classes=L, num_thing_classes=nt, num_stuff_classes=L - nt,
                          in_channels=256, out_channels=256, cat_stuff_mask=True, feat_downsample_stride=2, feat_refine=False,
                          use_binary=True, proposal_feats_with_obj=True, loss_seg=dict(type="FocalLoss", use_sigmoid=True),
                          loss_mask=dict(type="CrossEntropyLoss", use_sigmoid=True), localization_fpn=None))
    sd = Hh.seeded_fill({k: tuple(v.shape) for k, v in kh.state_dict().items()}, 77 + B)
    kh.load_state_dict(sd)
    kh.to(gpu)
    feats = Hh.neck_inputs(5 + B, B, 256, H, W)
    g = torch.Generator().manual_seed(B)
    names = ("proposal", "x", "mask_preds", "seg_preds", "dfe", "depth_pred")
    okeys = ("proposal_feats", "x_feats", "mask_preds", "seg_preds", "depth_feats", "depth_pred")
    with torch.enable_grad():
        w = {k_: v.clone().requires_grad_(True) for k_, v in sd.items()}
        cf = [f.clone().requires_grad_(True) for f in feats]
        r = O.kernel_head_post_neck(w, *cf, nt, L, 32, cat_stuff_mask=False)
        ref = [r[k_].reshape(B, 100, 256) if k_ == "proposal_feats" else r[k_] for k_ in okeys]
        cot = [torch.randn(t.shape, generator=g) * (0.05 if t.dim() == 4 else 1.0) for t in ref]
        sum((a * c).sum() for a, c in zip(ref, cot)).backward()
        df = [f.to(gpu).requires_grad_(True) for f in feats]
        for p in kh.parameters():
            p.grad = None
        out = T.rpn_forward(kh, df)
        sum((out[n] * c.to(gpu)).sum() for n, c in zip(names, cot)).backward()
    torch.cuda.synchronize()
    err = {n: Hh.rel_err(out[n].detach().cpu(), a.detach()) for n, a in zip(names, ref)}
    print("rpn node forward vs oracle:", {k_: f"{v:.1e}" for k_, v in err.items()})
    assert max(err.values()) < 2e-5, err
    gerr = {i: Hh.rel_err(df[i].grad.cpu(), cf[i].grad) for i in range(3)}
    perr = {n: Hh.rel_err(p.grad.cpu(), w[n].grad) for n, p in kh.named_parameters()}
    print("input gradients:", {k_: f"{v:.1e}" for k_, v in gerr.items()}, "parameter gradients:", {k_: f"{v:.1e}" for k_, v in perr.items()})
    assert max(gerr.values()) < 1e-4 and max(perr.values()) < 1e-4


def test_map_product_epilogues(gpu):
    """round 5 options of the two map products against torch: row bias, add source (also in place), row sums, batch sums"""
    g = torch.Generator().manual_seed(0)
    for B, M, K, H, W in ((2, 37, 256, 6, 10), (3, 160, 100, 5, 7)):
        A, X = torch.randn(B, M, K, generator=g).to(gpu), torch.randn(B, K, H, W, generator=g).to(gpu)
        bias, add = torch.randn(B, M, generator=g).to(gpu), torch.randn(B, M, H, W, generator=g).to(gpu)
        ref = torch.einsum("bmk,bkhw->bmhw", A.double(), X.double())
        y = T.rows_x_map(A, X, bias=bias, add=add)
        assert Hh.rel_err(y.cpu(), (ref + bias.double()[..., None, None] + add.double()).cpu()) < 2e-5
        y2 = add.clone()
        T.rows_x_map(A, X, out=y2, accumulate=True)
        assert Hh.rel_err(y2.cpu(), (ref + add.double()).cpu()) < 2e-5
        if K <= 256:
            Gm = torch.randn(B, M, H, W, generator=g).to(gpu)
            Xk = torch.randn(B, K, H, W, generator=g).to(gpu)
            rs = torch.empty((B, M), device=gpu)
            o = T.map_x_mapT(Gm, Xk, rowsum=rs)
            assert Hh.rel_err(o.cpu(), torch.einsum("bmhw,bkhw->bmk", Gm.double(), Xk.double()).cpu()) < 2e-5
            assert Hh.rel_err(rs.cpu(), Gm.double().sum((2, 3)).cpu()) < 1e-5
            rs1 = torch.empty((M,), device=gpu)
            o1 = T.map_x_mapT(Gm, Xk, rowsum=rs1, sum_batch=True)
            assert o1.shape == (M, K) and Hh.rel_err(o1.cpu(), torch.einsum("bmhw,bkhw->mk", Gm.double(), Xk.double()).cpu()) < 2e-5
            assert Hh.rel_err(rs1.cpu(), Gm.double().sum((0, 2, 3)).cpu()) < 1e-5
            cnt = torch.empty((B, M), device=gpu)
            T.map_x_mapT(Gm, Xk, binarize_g=True, rowsum=cnt)
            assert torch.equal(cnt.cpu(), (Gm > 1.5 * 2.0 ** -24).flatten(2).sum(-1).float().cpu())
