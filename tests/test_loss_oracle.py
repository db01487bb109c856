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


def test_rpn_forward_train_matches_reference():
    """KernelHead.forward_train (kernel_head.py:349-454) end to end in the oracle -- post-neck decode, x2 upsample, Hungarian
    assignment, rpn targets, rpn losses, depth_dense -- against the REFERENCE's own forward_train with its assigner, sampler
    and the real loss modules (tests/golden/train_rpn.npz)."""
    from oracle import assign_oracle as AO, poly_oracle as PO
    z = Hh.load_golden("train_rpn.npz")
    m = json.loads(bytes(z["meta_json"]).decode())
    B, H, W = m["B"], m["H"], m["W"]
    with open(Hh.GOLDEN + "/full_state_keys.json") as f:
        sd = Hh.seeded_fill(json.load(f), m["wseed"])
    nt, ns, Nq = 8, 11, 100
    L = nt + ns
    feats = Hh.neck_inputs(m["nseed"], B, 256, H, W)
    o = PO.kernel_head_post_neck(sd, *feats, nt, L, prefix="rpn_head.", cat_stuff_mask=False)
    up = lambda t: torch.nn.functional.interpolate(t, scale_factor=2, mode="bilinear", align_corners=False)
    smask, sseg, sdepth = up(o["mask_preds"]), up(o["seg_preds"]), up(o["depth_pred"])
    gts = [{k: torch.from_numpy(z[f"gt{b}_{k}"]) for k in ("masks", "labels", "sem_seg", "sem_cls", "depth")} for b in range(B)]
    valids = []
    for b, g in enumerate(gts):
        v = torch.cat((g["masks"], g["sem_seg"]), 0).sum(0).bool().float()
        g["gt_inds"], g["assigned_labels"] = AO.assign(smask[b], None, g["masks"], g["labels"], v)
        valids.append(v)
    tg = LO.rpn_get_targets(L, nt, ns, Nq, 2 * H, 2 * W, gts, valids)
    losses = LO.rpn_loss(L, smask, sseg, sdepth, *tg)
    gd = torch.stack([g["depth"][None] for g in gts])
    losses["depth_dense"] = LO.dense_depth(sdepth, gd)
    want = {k[2:]: float(np.asarray(z[k]).reshape(-1)[0]) for k in z.files if k.startswith("l_")}
    assert set(losses) == set(want)
    for k, v in losses.items():
        assert abs(float(v) - want[k]) <= 2e-5 * max(1.0, abs(want[k])), (k, float(v), want[k])
    # what forward_train hands to the roi head: stuff rows appended to masks / kernels (kernel_head.py:444-451)
    mp = torch.cat([o["mask_preds"], o["seg_preds"][:, nt:L]], 1)
    assert Hh.rel_err(mp, z["mask_preds"]) < 1e-5
    pf = torch.cat([o["proposal_feats"].reshape(B, Nq, 256), sd["rpn_head.conv_seg.weight"][nt:L].reshape(1, ns, 256).expand(B, ns, 256)], 1)
    assert Hh.rel_err(pf, z["proposal_feats"]) < 1e-5
