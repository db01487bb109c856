"""GPU debug: the training step's descriptor path against the generic (materialised targets) path on the train_step fixture:
per-stage loss values and d(losses)/d(predictions)"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as Hh  # noqa: E402
import test_gpu_loss as TL  # noqa: E402
from polyphonicformer_amd import train as T, losses as Lo  # noqa: E402
from polyphonicformer_amd.registry import HEADS  # noqa: E402
import polyphonicformer_amd.kernel_update  # noqa: F401,E402

torch.set_grad_enabled(True)
gpu = torch.device("cuda:0")
z = Hh.load_golden(sys.argv[1] if len(sys.argv) > 1 else "train_step.npz")
m = json.loads(bytes(z["meta_json"]).decode())
B, H, W, S = m["B"], m["H"], m["W"], m["S"]
rpn, sd = TL._rpn_head(gpu)
roi_a = dict(type='MaskHungarianAssignerWithDepth', cls_cost=dict(type='FocalLossCost', weight=2.0),
             dice_cost=dict(type='DiceCost', weight=4.0, pred_act=True), mask_cost=dict(type='MaskCost', weight=1.0, pred_act=True),
             depth_cost=dict(type='DepthCost', weight=0., loss_fn=dict(type='DepthMatchLoss', loss_weight=1.), depth_act_mode='sigmoid'))
roi = HEADS.build(dict(type="KernelUpdateIterHead", num_stages=S, assign_stages=S, stage_loss_weights=[1] * S, num_proposals=100,
                       num_thing_classes=8, num_stuff_classes=11, do_panoptic=True, merge_joint=True, mask_head=Hh.stage_cfg(256, 2048, 8, 19, 8, 11),
                       train_cfg=dict(assigner=roi_a, sampler=dict(type='MaskPseudoSampler'), pos_weight=1.)))
roi.load_state_dict({k[len("roi_head."):]: v for k, v in sd.items() if k.startswith("roi_head.")})
roi.to(gpu)
feats = [f.to(gpu) for f in Hh.neck_inputs(m["nseed"], B, 256, H, W)]
gts = [{k: torch.from_numpy(z[f"gt{b}_{k}"]).to(gpu) for k in ("masks", "labels", "sem_seg", "sem_cls", "depth")} for b in range(B)]
metas = [Hh.img_meta(H * 8, W * 8)] * B
gd = torch.stack([g["depth"][None] for g in gts])
gm, gl, gs, gc = [g["masks"] for g in gts], [g["labels"] for g in gts], [g["sem_seg"] for g in gts], [g["sem_cls"] for g in gts]
res = {}
real_ab = Lo.assign_batch
log = []
Lo.assign_batch = lambda *a: (log.append(real_ab(*a)), log[-1])[1]
for mode in ("fast", "generic"):
    T._fast_assign_ok = (lambda a, s: True) if mode == "fast" else (lambda a, s: False)
    r_losses, r = T.rpn_forward_train(rpn, feats, metas, gm, gl, gs, gc, gd, want_grads=True)
    k, mp, q = T.rpn_outputs(rpn, r)
    losses, _ = T.roi_forward_train(roi, r["x"], r["dfe"], k, mp, q, r["depth_pred"], metas, gm, gl, gs, gc, gd, want_grads=True)
    res[mode] = (r_losses, losses)
torch.cuda.synchronize()
fr, fl = res["fast"]
gr, gl_ = res["generic"]
for k_ in fl:
    if k_ == "_grads":
        continue
    print(k_, float(fl[k_]), float(gl_[k_]), "ref", float(z["l_" + k_]) if "l_" + k_ in z.files else None)
for s in range(S):
    for n in ("cls_score", "mask_pred", "depth_pred"):
        a, b = fl["_grads"][s][n], gl_["_grads"][s][n]
        print("stage", s, n, "rel diff", Hh.rel_err(a.cpu(), b.cpu()))
for n in ("mask_pred", "seg_preds", "depth_pred"):
    print("rpn", n, Hh.rel_err(fr["_grads"][n].cpu(), gr["_grads"][n].cpu()))
print("assignments (fast path):", [[(a.tolist(), b.tolist()) for a, b in st] for st in log])

# ---- hard decisions: the device's stage-input mask logits against a CPU replay with the oracle (free running)
from oracle import poly_oracle as O  # noqa: E402
with torch.no_grad():
    sdc = {k_: v for k_, v in sd.items()}
    kh = O.kernel_head_post_neck({k_[len("rpn_head."):]: v for k_, v in sdc.items() if k_.startswith("rpn_head.")},
                                 *[f.cpu() for f in feats], 8, 19, 32, cat_stuff_mask=True)
    r = T.rpn_forward(rpn, feats)
    k, mp, q = T.rpn_outputs(rpn, r)
    cm, ck, cq = kh["mask_preds"], kh["proposal_feats"].reshape(B, -1, 256), kh["depth_proposal"].reshape(B, -1, 256)
    x, dfe = r["x"], r["dfe"]
    m = mp
    roi_sd = {k_[len("roi_head."):]: v for k_, v in sdc.items() if k_.startswith("roi_head.")}
    for s in range(S):
        dz, cz = m.detach().cpu(), cm
        flips = ((dz > 0) != (cz > 0))
        small = cz.abs().flatten().topk(5, largest=False).values.tolist()
        print(f"stage {s} input masks: flips {int(flips.sum())} of {flips.numel()}, smallest |logit| (CPU): {['%.2e' % v for v in small]}, "
              f"max |device - cpu| {float((dz - cz).abs().max()):.2e}, at flips: {[(float(a), float(b)) for a, b in zip(dz[flips].tolist(), cz[flips].tolist())][:5]}")
        cls, m, k, depth, q = T.stage_forward(roi.mask_head[s], x, dfe, k, m, q)
        rr = O.update_stage(roi_sd, f"mask_head.{s}.", kh["x_feats"], ck, cm, cq, kh["depth_feats"])
        ck, cq, cm = rr["obj"], rr["dobj"], rr["mask"]
