"""CPU: the loss oracle (oracle/loss_oracle.py) against the reference's own KernelUpdateHead.get_targets / .loss with the
vendored mmdet losses and the project's DepthLoss (tests/golden/loss.npz, oracle/gen_golden_loss.py): targets bit for
bit, loss values to 1e-6, and the gradients w.r.t. the predictions (autograd through the restatement vs autograd
through the reference)."""
import json

import numpy as np
import pytest
import torch

import helpers as Hh
from oracle import loss_oracle as LO


def load_case(tag):
    z = Hh.load_golden("loss.npz")
    m = json.loads(bytes(z[f"{tag}_meta"]).decode())
    gts = [{k: torch.from_numpy(z[f"{tag}_gt{b}_{k}"]) for k in ("masks", "labels", "sem_seg", "sem_cls", "depth", "gt_inds", "assigned_labels")}
           for b in range(m["B"])]
    valids = [torch.from_numpy(z[f"{tag}_valid{b}"]) for b in range(m["B"])]
    return z, m, gts, valids


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_targets_and_losses_match_reference(tag):
    z, m, gts, valids = load_case(tag)
    L = m["n_thing"] + m["n_stuff"]
    tg = LO.get_targets(L, m["n_thing"], m["n_stuff"], m["Nq"], m["H"], m["W"], gts, valids)
    for k, t in zip(("labels", "label_weights", "mask_targets", "mask_weights", "depth_targets", "depth_weights"), tg):
        assert np.array_equal(t.numpy(), z[f"{tag}_t_{k}"]), k
    with torch.enable_grad():          # other test modules switch autograd off globally at import
        _check_losses(z, m, tag, tg, L)


def _check_losses(z, m, tag, tg, L):
    mp = torch.from_numpy(z[f"{tag}_mask_pred"]).requires_grad_(True)
    cs = torch.from_numpy(z[f"{tag}_cls_score"]).requires_grad_(True)
    dp = torch.from_numpy(z[f"{tag}_depth_pred"]).requires_grad_(True)
    losses = LO.stage_loss(L, cs, mp, dp, *tg)
    for k, v in losses.items():
        want = float(np.asarray(z[f"{tag}_l_{k}"]).reshape(-1)[0])
        assert abs(float(v.detach()) - want) <= 1e-6 * max(1.0, abs(want)), (k, float(v.detach()), want)
    sum(v for k, v in losses.items() if k.startswith("loss")).backward()
    for name, t in (("mask_pred", mp), ("cls_score", cs), ("depth_pred", dp)):
        assert Hh.rel_err(t.grad, z[f"{tag}_g_{name}"]) < 1e-5, name
