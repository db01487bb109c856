"""SURVEY.md 8(f) N3 -- the oracle of SemanticFPNWrapper against goldens produced by the reference class
(oracle/gen_golden_neck.py).  CPU only."""
import json
import os

import pytest
import torch

import helpers as Hh
from oracle import neck_oracle as NO

CASES = dict(mini=(32, 4, 16, 24, 40, 2), full=(256, 32, 128, 16, 32, 1))


def _state(tag):
    keys = json.load(open(os.path.join(Hh.GOLDEN, "neck_state_keys.json")))[tag]
    return Hh.seeded_fill({k: tuple(v) for k, v in keys.items()}, 31)


@pytest.mark.parametrize("tag", ["mini", "full"])
def test_neck_oracle_matches_reference(tag):
    C, groups, nf, H0, W0, B = CASES[tag]
    g = Hh.load_golden(f"{tag}_neck.npz")
    feats = Hh.fpn_inputs(seed=32, B=B, C=C, H0=H0, W0=W0)
    outs = NO.semantic_fpn(_state(tag), feats, groups=groups, num_feats=nf)
    for name, o in zip(("out", "aux0", "aux1"), outs):
        assert Hh.rel_err(o, torch.from_numpy(g[name])) < 2e-6, name
    pe = NO.sine_positional_encoding(B, H0 // 8, W0 // 8, nf)
    assert Hh.rel_err(pe, torch.from_numpy(g["posenc"])) < 1e-6


def test_neck_state_dict_layout():
    """the 7 3x3 towers + conv_pred + 2 aux convs of configs/_base_/models/polyphonic_former.py:78-96"""
    keys = json.load(open(os.path.join(Hh.GOLDEN, "neck_state_keys.json")))["full"]
    convs = sorted(k for k in keys if k.endswith("conv.weight"))
    assert len(convs) == 10
    assert keys["convs_all_levels.0.conv0.conv.weight"] == [256, 256, 3, 3]
    assert keys["conv_pred.conv.weight"] == [256, 256, 1, 1] and keys["aux_convs.1.gn.bias"] == [256]
