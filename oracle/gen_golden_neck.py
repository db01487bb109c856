"""Goldens for SURVEY.md 8(f) N3 (SemanticFPNWrapper), produced by the reference classes loaded from /root/reference.
    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_neck.py"""
import json, os, sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import helpers  # noqa: E402
from oracle import ref_loader, neck_oracle  # noqa: E402


def build(ns, nk, C, groups, num_feats, seed):
    m = nk.SemanticFPNWrapper(**ref_loader.neck_cfg(C, groups, num_feats))
    sd = m.state_dict()
    fill = helpers.seeded_fill({k: tuple(v.shape) for k, v in sd.items()}, seed)
    m.load_state_dict(fill)
    return m.eval(), fill


def main():
    ns = ref_loader.load_reference()
    nk = ref_loader.load_reference_neck()
    torch.manual_seed(0)
    keys = {}
    for tag, (C, groups, nf, H0, W0, B) in dict(mini=(32, 4, 16, 24, 40, 2), full=(256, 32, 128, 16, 32, 1)).items():
        m, sd = build(ns, nk, C, groups, nf, seed=31)
        feats = helpers.fpn_inputs(seed=32, B=B, C=C, H0=H0, W0=W0)
        with torch.no_grad():
            outs = m(feats)
            ours = neck_oracle.semantic_fpn(sd, feats, groups=groups, num_feats=nf)
            pe = nk.SinePositionalEncoding(num_feats=nf, normalize=True)(torch.zeros(B, H0 // 8, W0 // 8, dtype=torch.bool))
        err = max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(ours, outs))
        print(tag, "restatement vs reference: max rel err", err)
        assert err < 2e-6
        np.savez_compressed(os.path.join(REPO, "tests", "golden", f"{tag}_neck.npz"),
                            out=outs[0].numpy(), aux0=outs[1].numpy(), aux1=outs[2].numpy(), posenc=pe.numpy())
        keys[tag] = {k: list(v.shape) for k, v in m.state_dict().items()}
    json.dump(keys, open(os.path.join(REPO, "tests", "golden", "neck_state_keys.json"), "w"), indent=0)


def digest(t, n=256):
    """(norm, sum, n strided entries) of a gradient -- the same form as gen_golden_loss.grad_digest"""
    f = t.detach().double().reshape(-1)
    idx = torch.linspace(0, f.numel() - 1, n).long()
    return np.concatenate([[float(f.norm()), float(f.sum())], f[idx].numpy()])


def train_fixture(name="neck_train.npz", B=2, H0=16, W0=32, C=256, groups=32, nf=128):
    """round 5 (VERDICT r04 #1c): the reference SemanticFPNWrapper in TRAINING -- forward on four FPN levels, random cotangents on
    its three outputs, torch autograd backward -- pins the gradient of every neck parameter and of the four inputs.
    Stored: the outputs (strided), digests of every gradient; weights / inputs / cotangents are regenerated from seeds."""
    ns = ref_loader.load_reference()
    nk = ref_loader.load_reference_neck()
    m, sd = build(ns, nk, C, groups, nf, seed=33)
    m.train()
    feats = [f.requires_grad_(True) for f in helpers.fpn_inputs(seed=34, B=B, C=C, H0=H0, W0=W0)]
    g = torch.Generator().manual_seed(35)
    outs = m(feats)
    cots = [torch.randn(o.shape, generator=g) for o in outs]
    sum((o * c).sum() for o, c in zip(outs, cots)).backward()
    out = {"meta_json": np.frombuffer(json.dumps(dict(B=B, H0=H0, W0=W0, C=C, groups=groups, nf=nf, wseed=33, iseed=34, cseed=35)).encode(),
                                      dtype=np.uint8)}
    for j, o in enumerate(outs):
        out[f"out{j}"] = digest(o, 4096)
    for n, p in m.named_parameters():
        assert p.grad is not None, n
        out["g_" + n] = digest(p.grad)
    for i, f in enumerate(feats):
        out[f"gin{i}"] = digest(f.grad, 4096)
    np.savez_compressed(os.path.join(REPO, "tests", "golden", name), **out)
    print(name, "parameters:", sum(k.startswith("g_") for k in out), "output norms", [float(o.norm()) for o in outs])


if __name__ == "__main__":
    main()
    train_fixture()
    train_fixture("neck_train_b.npz", B=1, H0=24, W0=40)      # rows of 40 / 20 / 10 / 5 pixels: no 16-byte alignment at the coarse levels
