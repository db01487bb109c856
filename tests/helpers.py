"""Deterministic, platform-independent test inputs shared by `oracle/gen_golden.py` (build container,
reference available) and the tests (both boxes).  Everything is drawn from a seeded CPU
`torch.Generator`, so the GPU box can regenerate the exact weights / feature maps a golden fixture
was produced from without the fixture having to store them."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, "tests", "golden")
if REPO not in sys.path:
    sys.path.insert(0, REPO)

# the shipped Cityscapes config (configs/_base_/models/polyphonic_former.py:1-6,111-126)
FULL = dict(C=256, F=2048, heads=8, groups=32, n_thing=8, n_stuff=11, Nq=100, S=3)
# reduced dims for fast oracle-vs-reference checks (the reference classes are parametric)
MINI = dict(C=32, F=64, heads=4, groups=4, n_thing=3, n_stuff=2, Nq=7, S=2)


def seeded_fill(shapes, seed):
    """shapes: {key: shape}.  Returns {key: fp32 tensor}; keys are visited in sorted order and all
    values come from ONE generator, so the result depends only on (key set, shapes, seed)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k in sorted(shapes):
        shp = tuple(shapes[k])
        t = torch.randn(shp, generator=g)
        if len(shp) >= 2:                       # linear / conv weights: xavier-like scale
            fan_out = shp[0]
            fan_in = int(np.prod(shp[1:]))
            t = t * float(np.sqrt(2.0 / (fan_in + fan_out)))
            if k.endswith("init_kernels.weight"):
                t = t * 4.0                     # kernels need O(1) logits to give non-trivial masks
        elif k.endswith("weight"):              # LayerNorm / GroupNorm gains
            t = 1.0 + 0.1 * t
        else:                                   # biases
            t = 0.1 * t
        out[k] = t.contiguous()
    return out


def iter_inputs(seed, B, N, C, H, W, mask_bias=0.0):
    """Isolated-IterHead inputs as BASELINE.md section 2 defines them."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, C, H, W, generator=g)
    dfe = torch.randn(B, C, H, W, generator=g)
    k0 = torch.randn(B, N, C, 1, 1, generator=g)
    dker = torch.randn(1, 1, C, 1, 1, generator=g)
    q0 = dker.expand(B, N, C, 1, 1)           # stride-0 view, like kernel_head.py:336
    m0 = torch.randn(B, N, H, W, generator=g) + mask_bias
    dpr = torch.randn(B, 1, H, W, generator=g)
    return dict(x=x, dfe=dfe, k0=k0, q0=q0, m0=m0, depth_pred=dpr)


def neck_inputs(seed, B, C, H, W):
    """Post-neck maps ~ ReLU(N(0,1)) (they follow a ReLU in semantic_fpn.py:158-178)."""
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(B, C, H, W, generator=g).relu() for _ in range(3)]


def img_meta(h, w, pad_to=None, ori=None):
    bh, bw = pad_to if pad_to else (h, w)
    oh, ow = ori if ori else (h, w)
    return dict(img_shape=(h, w, 3), ori_shape=(oh, ow, 3), batch_input_shape=(bh, bw))


def rel_err(a, b):
    """max |a-b| / max|b| -- the '1e-3 relative' of BASELINE.json:north_star."""
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def needed_atol(a, b, rtol):
    """the smallest `atol`, as a fraction of max|b|, for which |a - b| <= rtol * |b| + atol * max|b| holds on EVERY element
    (the mixed relative / absolute criterion of numpy.allclose, with the absolute part scaled by the tensor's magnitude).
    Asserted below the max-normalised tolerance it bounds the error of the small entries more tightly than `rel_err` does
    and lets a large entry use its own magnitude: it can fail where `rel_err` passes."""
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    d = (a - b).abs() - rtol * b.abs()
    return float(d.max().clamp_min(0.0) / b.abs().max().clamp_min(1e-30))


def unpack_hard_masks(dev_bits, N, H, W):
    """`DecodePlan.debug_bits` (the bit words [B, Npad, HWp / 32] every stage of a device run pooled with) as {0, 1}
    fp32 tensors [B, N, H, W]: what the oracle takes as `hard_masks=` so that a free-running comparison follows the
    DEVICE's hard decisions and measures arithmetic at 1e-3, whatever pixel flipped"""
    out = []
    for b in dev_bits:
        u = np.unpackbits(b.cpu().numpy().view("uint32").view("uint8"), axis=-1, bitorder="little")[:, :N, :H * W]
        out.append(torch.from_numpy(u.astype("float32")).reshape(b.shape[0], N, H, W))
    return out


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def tracker_records(seed, nframes=12, nobj=14):
    """a synthetic clip: objects drift, appear and disappear; embeddings = identity vector + noise"""
    g = torch.Generator().manual_seed(seed)
    ident = torch.randn(nobj, 256, generator=g) * 0.5
    pos = torch.rand(nobj, 2, generator=g) * torch.tensor([1800.0, 800.0])
    size = 40 + torch.rand(nobj, 2, generator=g) * 200
    cls = torch.randint(0, 8, (nobj,), generator=g)
    recs = []
    for f in range(nframes):
        alive = torch.rand(nobj, generator=g) < 0.8
        pos = pos + torch.randn(nobj, 2, generator=g) * 8
        idx = alive.nonzero().squeeze(1)
        idx = idx[torch.randperm(len(idx), generator=g)]
        bb = torch.cat([pos[idx], pos[idx] + size[idx], 0.2 + 0.8 * torch.rand(len(idx), 1, generator=g)], 1)
        emb = ident[idx] + 0.15 * torch.randn(len(idx), 256, generator=g)
        recs.append((f, bb, cls[idx].clone(), emb))
    return recs


def video_case(seed=11, H=96, W=160, nseg=9):
    """a crafted panoptic id map with `nseg` thing blobs (ids 1..nseg) + per-segment labels/scores, 4 FPN levels of
    random features for a (8H x 8W... here H x W image) and random RoI features -- inputs of the association step"""
    g = torch.Generator().manual_seed(seed)
    pan = np.zeros((H, W), dtype=np.int32)
    ys, xs = np.mgrid[0:H, 0:W]
    info = []
    for s in range(1, nseg + 1):
        cy, cx = float(torch.rand(1, generator=g)) * H, float(torch.rand(1, generator=g)) * W
        ry, rx = 3 + float(torch.rand(1, generator=g)) * H / 5, 3 + float(torch.rand(1, generator=g)) * W / 5
        m = ((ys - cy) / ry) ** 2 + ((xs - cx) / rx) ** 2 < 1
        if m.sum() == 0:
            m[int(min(cy, H - 1)), int(min(cx, W - 1))] = True
        pan[m] = s
    for s in range(1, nseg + 1):
        if (pan == s).any():
            info.append(dict(id=s, isthing=True, category_id=int(torch.randint(0, 8, (1,), generator=g)),
                             score=float(0.3 + 0.7 * torch.rand(1, generator=g)), instance_id=s - 1))
    feats = [torch.randn(1, 256, max(H // st, 1), max(W // st, 1), generator=g) for st in (4, 8, 16, 32)]
    roi_feats = torch.randn(len(info), 256, 7, 7, generator=g)
    return pan, info, feats, roi_feats


TRACK_HEAD_SHAPES = {**{f"track_head.convs.{i}.conv.weight": (256, 256, 3, 3) for i in range(4)},
                     **{f"track_head.convs.{i}.gn.weight": (256,) for i in range(4)},
                     **{f"track_head.convs.{i}.gn.bias": (256,) for i in range(4)},
                     "track_head.fcs.0.weight": (1024, 12544), "track_head.fcs.0.bias": (1024,),
                     "track_head.fc_embed.weight": (256, 1024), "track_head.fc_embed.bias": (256,)}


def dvps_clip(seed, nseq=2, nframes=5, H=32, W=64, num_classes=19, num_things=8):
    """synthetic DVPS evaluation data: per frame gt (sem, instance, depth) and a perturbed prediction
    (sem, track id, depth) -- blobs that drift, label noise, an ignore region (class 255), depth holes"""
    rng = np.random.default_rng(seed)
    ys, xs = np.mgrid[0:H, 0:W]
    frames = []
    for s in range(nseq):
        nobj = 6
        cy, cx = rng.uniform(0, H, nobj), rng.uniform(0, W, nobj)
        ry, rx = rng.uniform(3, H / 3, nobj), rng.uniform(3, W / 4, nobj)
        cls = rng.integers(0, num_things, nobj)
        stuff = rng.integers(num_things, num_classes, 3)
        for f in range(nframes):
            cy = cy + rng.normal(0, 1.0, nobj)
            cx = cx + rng.normal(0, 1.5, nobj)
            sem = np.where(ys < H // 3, stuff[0], np.where(xs < W // 2, stuff[1], stuff[2])).astype(np.int64)
            ins = np.zeros((H, W), dtype=np.int64)
            psem, ptrk = sem.copy(), ins.copy()
            for o in range(nobj):
                m = ((ys - cy[o]) / ry[o]) ** 2 + ((xs - cx[o]) / rx[o]) ** 2 < 1
                sem[m], ins[m] = cls[o], o + 1
                mp = ((ys - cy[o] - rng.normal(0, 0.7)) / (ry[o] * rng.uniform(0.8, 1.2))) ** 2 + \
                     ((xs - cx[o] - rng.normal(0, 0.7)) / rx[o]) ** 2 < 1
                wrong = rng.random() < 0.15
                psem[mp] = rng.integers(0, num_things) if wrong else cls[o]
                ptrk[mp] = (o + 1) if rng.random() > 0.1 else o + 20
            ign = (ys > H - 4) & (xs > W - 10)
            sem[ign], ins[ign] = 255, 0
            depth = (5 + 0.5 * ys + 0.1 * xs + rng.normal(0, 0.2, (H, W))).astype(np.float32)
            depth[rng.random((H, W)) < 0.05] = 0.
            pdepth = (np.maximum(depth, 1.0) * (1 + rng.normal(0, 0.08, (H, W)))).astype(np.float32)
            frames.append(dict(seq=s + 3, img=f * 5, gt=dict(sem=sem, track=ins, depth=depth),
                               pred=dict(sem=psem, track=ptrk, depth=pdepth)))
    return frames


def fpn_inputs(seed, B, C, H0, W0):
    """4 FPN levels (strides 4, 8, 16, 32 of an 4*H0 x 4*W0 image) ~ N(0, 1)"""
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(B, C, H0 >> i, W0 >> i, generator=g) for i in range(4)]


def assign_case(seed, N, G, L, H, W, with_valid=True, with_cls=True):
    """one image of the training-time assignment: mask logits [N,H,W], class logits [N,L], soft ground-
    truth masks [G,H,W] (binary blobs averaged over 2x2, i.e. what the x4 bilinear downsample of polyphonic_former.py:77-80
    produces: values k/4), labels [G], valid [H,W] of 0/1.  Half of the predictions are noisy copies of a gt mask, so the
    assignment is decided by the mask costs, the rest is noise."""
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(2 * H).float(), torch.arange(2 * W).float(), indexing="ij")
    gts = []
    for _ in range(G):
        cy, cx = torch.rand(1, generator=g) * 2 * H, torch.rand(1, generator=g) * 2 * W
        ry, rx = 2 + torch.rand(1, generator=g) * H * 0.6, 2 + torch.rand(1, generator=g) * W * 0.6
        gts.append((((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1).float())
    gt_hi = torch.stack(gts) if G else torch.zeros(0, 2 * H, 2 * W)
    gt = torch.nn.functional.avg_pool2d(gt_hi[None], 2)[0] if G else torch.zeros(0, H, W)
    logits = torch.randn(N, H, W, generator=g) * 2
    for n in range(0, N, 2):
        if G:
            k = int(torch.randint(0, G, (1,), generator=g))
            logits[n] = (gt[k] - 0.5) * 6 + torch.randn(H, W, generator=g)
    cls = torch.randn(N, L, generator=g) if with_cls else None
    labels = torch.randint(0, L, (G,), generator=g)
    valid = (torch.rand(H, W, generator=g) > 0.15).float() if with_valid else None
    return dict(mask_logits=logits, cls_logits=cls, gt_masks=gt, gt_labels=labels, gt_valid=valid)


def assign_depth_inputs(seed, N, H, W):
    """depth logits [N, H, W] and a ground-truth depth map [1, H, W] (10 % unlabelled = 0) for the DepthCost cases"""
    g = torch.Generator().manual_seed(seed + 1000)
    z = torch.randn(N, H, W, generator=g)
    d = torch.rand(1, H, W, generator=g) * 70.0 + 1.0
    d[torch.rand(1, H, W, generator=g) < 0.1] = 0.0
    return z, d


DEPTH_COST_CASES = [0, 2]          # indices into ASSIGN_CASES that also have a golden with DepthCost weight 1

ASSIGN_CASES = [dict(seed=11, N=100, G=7, L=8, H=24, W=40, with_valid=True),
                dict(seed=12, N=100, G=33, L=8, H=16, W=24, with_valid=True),
                dict(seed=13, N=37, G=3, L=8, H=7, W=13, with_valid=False),
                dict(seed=14, N=200, G=64, L=8, H=12, W=20, with_valid=True),
                dict(seed=15, N=20, G=1, L=8, H=8, W=8, with_valid=True, with_cls=False)]


def stage_cfg(C=256, F=2048, heads=8, L=19, n_thing=8, n_stuff=11):
    """`mask_head` config dict with the shipped structure (configs/_base_/models/polyphonic_former.py:111-165) at
    parametric width; the same dict `oracle/ref_loader.py` builds the reference heads from when it generates goldens"""
    return dict(
        type="KernelUpdateHead", num_thing_classes=n_thing, num_stuff_classes=n_stuff,
        num_classes=L, num_ffn_fcs=2, num_heads=heads, num_cls_fcs=1, num_mask_fcs=1,
        feedforward_channels=F, in_channels=C, out_channels=C, dropout=0.0, mask_thr=0.5,
        conv_kernel_size=1, mask_upsample_stride=2, ffn_act_cfg=dict(type="ReLU", inplace=True),
        with_ffn=True, feat_transform_cfg=dict(conv_cfg=dict(type="Conv2d"), act_cfg=None),
        kernel_updator_cfg=dict(type="KernelUpdator", in_channels=C, feat_channels=C,
                                out_channels=C, input_feat_shape=3,
                                act_cfg=dict(type="ReLU", inplace=True), norm_cfg=dict(type="LN")),
        loss_rank=dict(type="CrossEntropyLoss", use_sigmoid=False, loss_weight=0.1),
        loss_mask=dict(type="CrossEntropyLoss", use_sigmoid=True, loss_weight=1.0),
        loss_dice=dict(type="DiceLoss", loss_weight=4.0),
        loss_cls=dict(type="FocalLoss", use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=2.0),
        loss_depth=dict(type="DepthLoss", loss_weight=5.0, depth_act_mode="sigmoid"),
        depth_act_mode="sigmoid")


def train_gt(seed, B, H, W, n_thing, n_stuff, gts):
    """ground truth of a training step at the assign stride (H x W = the x2-upsampled mask size), shared with the tests"""
    g = torch.Generator().manual_seed(seed)
    out = []
    for b in range(B):
        G = gts[b]
        masks = (torch.rand(G, H, W, generator=g) > 0.75).float()
        labels = torch.randint(0, n_thing, (G,), generator=g)
        present = torch.randperm(n_stuff, generator=g)[: n_stuff // 2 + 1].sort()[0]
        sem_cls = present + n_thing
        sem_seg = (torch.rand(len(present), H, W, generator=g) > 0.6).float()
        depth = torch.rand(H, W, generator=g) * 90.0
        depth[torch.rand(H, W, generator=g) < 0.1] = 0.0
        out.append(dict(masks=masks, labels=labels, sem_seg=sem_seg, sem_cls=sem_cls, depth=depth))
    return out
