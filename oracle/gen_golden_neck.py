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


if __name__ == "__main__":
    main()
