"""TEST INFRASTRUCTURE ONLY -- loader that imports the *reference* hot-path files in this container.

The reference (`/root/reference`, read-only) hard-requires mmcv-full 1.3.18 and mmdet, neither of
which is installed here or on the GPU box.  This module pre-seeds ``sys.modules`` with minimal
stand-ins for the ~25 mmcv/mmdet symbols the three hot-path files touch (SURVEY.md section 8c and
Appendix C list them) and then loads, *by file path*,

    polyphonic/funcs/{depth_utils,utils,sampler,kernel_updator}.py
    polyphonic/{kernel_update_head,kernel_update,kernel_head}.py

so that ``oracle/gen_golden.py`` can run the real reference code on CPU and write golden vectors
into ``tests/golden``.  Nothing here is copied from the reference; the stand-ins re-state the
*published semantics* of the mmcv 1.3.18 bricks as thin compositions of ``torch.nn`` primitives:

* ``ConvModule``  = Conv2d (bias iff no norm, "auto") -> norm (attribute named ``gn``/``bn``) -> ReLU
  unless ``act_cfg=None``.
* ``FFN``         = ``x + Linear(ReLU(Linear(x)))`` with ``layers = Sequential(Sequential(Linear,
  act, Dropout), Linear, Dropout)``; legacy ``dropout=`` kwarg accepted.
* ``MultiheadAttention`` = ``identity + nn.MultiheadAttention(q=k=v=x)[0]`` (sequence first),
  attribute ``attn``.
* ``build_norm_layer(dict(type='LN'|'GN'), C)`` -> ``(name, nn.LayerNorm|nn.GroupNorm)`` with
  eps 1e-5.

This file never travels to the GPU box in any useful form: ``/root/reference`` does not exist
there, and nothing in the product path, the ``-m gpu`` tests, ``smoke()`` or ``bench.py`` imports it.
"""
import copy
import importlib.util
import os
import sys
import types

import torch
import torch.nn as nn

REF_ROOT = os.environ.get("POLY_REFERENCE_ROOT", "/root/reference")


class ConfigDict(dict):
    """dict with nested attribute access (what mmcv's addict-based ConfigDict offers)."""

    def __getattr__(self, name):
        try:
            v = self[name]
        except KeyError:
            raise AttributeError(name)
        if isinstance(v, dict) and not isinstance(v, ConfigDict):
            v = ConfigDict(v)
            self[name] = v
        return v

    def __setattr__(self, name, value):
        self[name] = value


class Registry:
    def __init__(self, name):
        self.name = name
        self._module_dict = {}

    def register_module(self, name=None, force=False, module=None):
        if module is not None:
            self._module_dict[name or module.__name__] = module
            return module

        def _reg(cls):
            self._module_dict[name or cls.__name__] = cls
            return cls

        return _reg

    def get(self, key):
        return self._module_dict.get(key)

    def build(self, cfg, default_args=None):
        cfg = dict(cfg)
        if default_args:
            for k, v in default_args.items():
                cfg.setdefault(k, v)
        typ = cfg.pop("type")
        cls = self._module_dict[typ] if isinstance(typ, str) else typ
        return cls(**cfg)


def _mod(name):
    m = types.ModuleType(name)
    m.__path__ = []  # behave as a package
    sys.modules[name] = m
    parent, _, child = name.rpartition(".")
    if parent and parent in sys.modules:
        setattr(sys.modules[parent], child, m)
    return m


def build_norm_layer(cfg, num_features, postfix=""):
    cfg = dict(cfg)
    typ = cfg.pop("type")
    requires_grad = cfg.pop("requires_grad", True)
    cfg.setdefault("eps", 1e-5)
    if typ == "LN":
        layer, abbr = nn.LayerNorm(num_features, **cfg), "ln"
    elif typ == "GN":
        layer, abbr = nn.GroupNorm(num_channels=num_features, **cfg), "gn"
    elif typ == "BN":
        layer, abbr = nn.BatchNorm2d(num_features, **cfg), "bn"
    else:
        raise KeyError(typ)
    for p in layer.parameters():
        p.requires_grad = requires_grad
    return abbr + str(postfix), layer


def build_activation_layer(cfg):
    cfg = dict(cfg)
    typ = cfg.pop("type")
    return {"ReLU": nn.ReLU, "GELU": nn.GELU, "Sigmoid": nn.Sigmoid}[typ](**cfg)


class ConvModule(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias="auto", conv_cfg=None, norm_cfg=None,
                 act_cfg=dict(type="ReLU"), inplace=True, **kw):
        super().__init__()
        self.with_norm = norm_cfg is not None
        self.with_activation = act_cfg is not None
        if bias == "auto":
            bias = not self.with_norm
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride,
                              padding=padding, dilation=dilation, groups=groups, bias=bias)
        if self.with_norm:
            self.norm_name, norm = build_norm_layer(norm_cfg, out_channels)
            self.add_module(self.norm_name, norm)
        if self.with_activation:
            act_cfg = dict(act_cfg)
            if act_cfg["type"] == "ReLU":
                act_cfg.setdefault("inplace", inplace)
            self.activate = build_activation_layer(act_cfg)

    def forward(self, x):
        x = self.conv(x)
        if self.with_norm:
            x = getattr(self, self.norm_name)(x)
        if self.with_activation:
            x = self.activate(x)
        return x


class FFN(nn.Module):
    def __init__(self, embed_dims=256, feedforward_channels=1024, num_fcs=2,
                 act_cfg=dict(type="ReLU", inplace=True), ffn_drop=0.0, dropout_layer=None,
                 add_identity=True, init_cfg=None, **kwargs):
        super().__init__()
        if "dropout" in kwargs:
            ffn_drop = kwargs["dropout"]
        layers = []
        in_c = embed_dims
        for _ in range(num_fcs - 1):
            layers.append(nn.Sequential(nn.Linear(in_c, feedforward_channels),
                                        build_activation_layer(act_cfg), nn.Dropout(ffn_drop)))
            in_c = feedforward_channels
        layers.append(nn.Linear(feedforward_channels, embed_dims))
        layers.append(nn.Dropout(ffn_drop))
        self.layers = nn.Sequential(*layers)
        self.add_identity = add_identity

    def forward(self, x, identity=None):
        out = self.layers(x)
        if not self.add_identity:
            return out
        return (x if identity is None else identity) + out


class MultiheadAttention(nn.Module):
    def __init__(self, embed_dims, num_heads, attn_drop=0.0, proj_drop=0.0, dropout_layer=None,
                 init_cfg=None, batch_first=False, **kwargs):
        super().__init__()
        if "dropout" in kwargs:
            attn_drop = kwargs["dropout"]
        self.attn = nn.MultiheadAttention(embed_dims, num_heads, attn_drop)
        self.proj_drop = nn.Dropout(proj_drop)

    def forward(self, query, key=None, value=None, identity=None, **kw):
        key = query if key is None else key
        value = key if value is None else value
        identity = query if identity is None else identity
        out = self.attn(query=query, key=key, value=value)[0]
        return identity + self.proj_drop(out)


class _Loss(nn.Module):
    def __init__(self, use_sigmoid=False, **kw):
        super().__init__()
        self.use_sigmoid = use_sigmoid


class IdentityNeck(nn.Module):
    """Stand-in for `localization_fpn`: passes the three post-neck maps through (the
    hot path starts *after* `localization_fpn(img)`, kernel_head.py:243)."""

    def __init__(self, **kw):
        super().__init__()

    def init_weights(self):
        pass

    def forward(self, x):
        return list(x)


_LOADED = None


def load_reference():
    """Returns a namespace with the reference classes KernelHead, KernelUpdateIterHead,
    KernelUpdateHead, KernelUpdator and the registry used to build them."""
    global _LOADED
    if _LOADED is not None:
        return _LOADED
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError(f"reference tree not found at {REF_ROOT}")
    sys.dont_write_bytecode = True

    MODELS = Registry("models")
    TRANSFORMER_LAYER = Registry("transformer_layer")
    POSITIONAL_ENCODING = Registry("pos_enc")
    BBOX_ASSIGNERS = Registry("assigner")
    BBOX_SAMPLERS = Registry("sampler")
    for n in ("FocalLoss", "CrossEntropyLoss", "DiceLoss", "DepthLoss"):
        MODELS.register_module(name=n, module=type(n, (_Loss,), {}))
    MODELS.register_module(name="IdentityNeck", module=IdentityNeck)

    mmcv = _mod("mmcv")
    cnn = _mod("mmcv.cnn")
    cnn.ConvModule = ConvModule
    cnn.build_norm_layer = build_norm_layer
    cnn.build_activation_layer = build_activation_layer
    cnn.bias_init_with_prob = lambda p: float(-torch.log(torch.tensor((1 - p) / p)))

    def normal_init(module, mean=0, std=1, bias=0):
        nn.init.normal_(module.weight, mean, std)
        if getattr(module, "bias", None) is not None:
            nn.init.constant_(module.bias, bias)

    cnn.normal_init = normal_init
    _mod("mmcv.cnn.bricks")
    tr = _mod("mmcv.cnn.bricks.transformer")
    tr.FFN, tr.MultiheadAttention = FFN, MultiheadAttention
    tr.TRANSFORMER_LAYER = TRANSFORMER_LAYER
    tr.POSITIONAL_ENCODING = POSITIONAL_ENCODING
    tr.build_transformer_layer = lambda cfg, default_args=None: TRANSFORMER_LAYER.build(cfg, default_args)
    tr.build_positional_encoding = lambda cfg, default_args=None: POSITIONAL_ENCODING.build(cfg, default_args)
    runner = _mod("mmcv.runner")
    class BaseModule(nn.Module):          # mmcv.runner.BaseModule: nn.Module + an ignored init_cfg
        def __init__(self, init_cfg=None):
            super().__init__()

    runner.BaseModule = BaseModule

    def force_fp32(apply_to=None, out_fp16=False):
        return lambda f: f

    runner.force_fp32 = force_fp32
    utils = _mod("mmcv.utils")
    utils.Registry = Registry
    ops = _mod("mmcv.ops")
    ops.DeformConv2dPack = type("DeformConv2dPack", (nn.Module,), {})

    _mod("mmdet")
    core = _mod("mmdet.core")
    core.build_assigner = lambda cfg, **kw: BBOX_ASSIGNERS.build(cfg)
    core.build_sampler = lambda cfg, **kw: BBOX_SAMPLERS.build(cfg)

    def multi_apply(func, *args, **kwargs):
        from functools import partial
        pfunc = partial(func, **kwargs) if kwargs else func
        return tuple(map(list, zip(*map(pfunc, *args))))

    core.multi_apply = multi_apply
    core.reduce_mean = lambda t: t
    bbox = _mod("mmdet.core.bbox")
    bbox.BaseSampler = type("BaseSampler", (), {"__init__": lambda self, *a, **k: None})
    bbox.SamplingResult = type("SamplingResult", (), {})
    bb = _mod("mmdet.core.bbox.builder")
    bb.BBOX_ASSIGNERS, bb.BBOX_SAMPLERS = BBOX_ASSIGNERS, BBOX_SAMPLERS
    _mod("mmdet.models")
    mb = _mod("mmdet.models.builder")
    for n in ("HEADS", "NECKS", "LOSSES", "BACKBONES", "DETECTORS", "ROI_EXTRACTORS", "SHARED_HEADS"):
        setattr(mb, n, MODELS)
    mb.build_loss = mb.build_neck = mb.build_head = lambda cfg: MODELS.build(cfg)
    losses = _mod("mmdet.models.losses")
    losses.accuracy = lambda *a, **k: None
    rh = _mod("mmdet.models.roi_heads")

    class BaseRoIHead(nn.Module):
        # call order follows mmdet/models/roi_heads/base_roi_head.py:32-35
        def __init__(self, bbox_roi_extractor=None, bbox_head=None, mask_roi_extractor=None,
                     mask_head=None, shared_head=None, train_cfg=None, test_cfg=None,
                     pretrained=None, init_cfg=None):
            super().__init__()
            self.train_cfg = train_cfg
            self.test_cfg = test_cfg
            if mask_head is not None:
                self.init_mask_head(mask_roi_extractor, mask_head)
            self.init_assigner_sampler()

    rh.BaseRoIHead = BaseRoIHead
    _mod("mmdet.models.dense_heads")
    ah = _mod("mmdet.models.dense_heads.atss_head")
    ah.reduce_mean = lambda t: t
    mu = _mod("mmdet.utils")
    import logging
    mu.get_root_logger = lambda *a, **k: logging.getLogger("ref")

    _mod("polyphonic")
    _mod("polyphonic.funcs")

    def load(modname, relpath):
        spec = importlib.util.spec_from_file_location(modname, os.path.join(REF_ROOT, relpath))
        m = importlib.util.module_from_spec(spec)
        sys.modules[modname] = m
        parent, _, child = modname.rpartition(".")
        setattr(sys.modules[parent], child, m)
        spec.loader.exec_module(m)
        return m

    load("polyphonic.funcs.depth_utils", "polyphonic/funcs/depth_utils.py")
    load("polyphonic.funcs.utils", "polyphonic/funcs/utils.py")
    load("polyphonic.funcs.sampler", "polyphonic/funcs/sampler.py")
    ku = load("polyphonic.funcs.kernel_updator", "polyphonic/funcs/kernel_updator.py")
    kuh = load("polyphonic.kernel_update_head", "polyphonic/kernel_update_head.py")
    kui = load("polyphonic.kernel_update", "polyphonic/kernel_update.py")
    kh = load("polyphonic.kernel_head", "polyphonic/kernel_head.py")

    ns = types.SimpleNamespace(
        MODELS=MODELS, TRANSFORMER_LAYER=TRANSFORMER_LAYER, ConfigDict=ConfigDict,
        KernelUpdator=ku.KernelUpdator, KernelUpdateHead=kuh.KernelUpdateHead,
        KernelUpdateIterHead=kui.KernelUpdateIterHead, KernelHead=kh.KernelHead,
        depth_act=sys.modules["polyphonic.funcs.depth_utils"].depth_act)
    _LOADED = ns
    return ns


def load_reference_tracker():
    """the reference's QuasiDenseEmbedTracker class (needs only mmdet.core.bbox_overlaps and a TRACKERS registry)"""
    load_reference()
    core = sys.modules["mmdet.core"]
    if not hasattr(core, "bbox_overlaps"):
        def bbox_overlaps(b1, b2, mode="iou", is_aligned=False, eps=1e-6):
            # mmdet/core/bbox/iou_calculators/iou2d_calculator.py semantics for mode='iou', is_aligned=False
            area1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
            area2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
            if b1.shape[0] == 0 or b2.shape[0] == 0:
                return b1.new_zeros((b1.shape[0], b2.shape[0]))
            lt = torch.max(b1[:, None, :2], b2[None, :, :2])
            rb = torch.min(b1[:, None, 2:], b2[None, :, 2:])
            wh = (rb - lt).clamp(min=0)
            overlap = wh[..., 0] * wh[..., 1]
            union = torch.max(area1[:, None] + area2[None, :] - overlap, overlap.new_tensor([eps]))
            return overlap / union
        core.bbox_overlaps = bbox_overlaps
    for name in ("polyphonic.video", "polyphonic.video.qdtrack", "polyphonic.video.qdtrack.trackers"):
        if name not in sys.modules:
            _mod(name)
    b = _mod("polyphonic.video.qdtrack.builder")
    b.TRACKERS = Registry("trackers")
    spec = importlib.util.spec_from_file_location(
        "polyphonic.video.qdtrack.trackers.quasi_dense_embed_tracker",
        os.path.join(REF_ROOT, "polyphonic/video/qdtrack/trackers/quasi_dense_embed_tracker.py"))
    m = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = m
    spec.loader.exec_module(m)
    return m.QuasiDenseEmbedTracker


def load_reference_track_head():
    """QuasiDenseMaskEmbedHeadGTMask + the mask->box helpers of polyphonic/video/utils.py"""
    ns = load_reference()
    for n in ("MultiPosCrossEntropyLoss", "L2Loss"):
        ns.MODELS.register_module(name=n, module=type(n, (_Loss,), {}))
    for name in ("polyphonic.video", "polyphonic.video.qdtrack", "polyphonic.video.qdtrack.track"):
        if name not in sys.modules:
            _mod(name)

    def load(modname, relpath):
        spec = importlib.util.spec_from_file_location(modname, os.path.join(REF_ROOT, relpath))
        m = importlib.util.module_from_spec(spec)
        sys.modules[modname] = m
        parent, _, child = modname.rpartition(".")
        setattr(sys.modules[parent], child, m)
        spec.loader.exec_module(m)
        return m

    sim = load("polyphonic.video.qdtrack.track.similarity", "polyphonic/video/qdtrack/track/similarity.py")
    sys.modules["polyphonic.video.qdtrack.track"].cal_similarity = sim.cal_similarity
    vu = load("polyphonic.video.utils", "polyphonic/video/utils.py")
    th = load("polyphonic.video.track_heads", "polyphonic/video/track_heads.py")
    return types.SimpleNamespace(Head=th.QuasiDenseMaskEmbedHeadGTMask, batch_mask2boxlist=vu.batch_mask2boxlist,
                                 bboxlist2roi=vu.bboxlist2roi,
                                 tensor_mask2box=sys.modules["polyphonic.funcs.utils"].tensor_mask2box)


def stage_cfg(C=256, F=2048, heads=8, L=19, n_thing=8, n_stuff=11):
    """`mask_head` dict with the shipped config's structure (configs/_base_/models/
    polyphonic_former.py:111-165) at parametric width."""
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


def build_iter_head(ns, S=3, N_thing_q=100, C=256, F=2048, heads=8, n_thing=8, n_stuff=11,
                    test_cfg=None):
    L = n_thing + n_stuff
    if test_cfg is None:
        test_cfg = ConfigDict(max_per_img=N_thing_q, mask_thr=0.5, stuff_score_thr=0.05,
                              merge_stuff_thing=dict(overlap_thr=0.6, iou_thr=0.5,
                                                     stuff_max_area=4096, instance_score_thr=0.3))
    head = ns.KernelUpdateIterHead(
        num_stages=S, assign_stages=S, recursive=False, stage_loss_weights=[1] * S,
        proposal_feature_channel=C, num_proposals=N_thing_q, num_thing_classes=n_thing,
        num_stuff_classes=n_stuff, do_panoptic=True, merge_joint=True,
        # each stage gets its own deep copy: feat_transform_cfg.pop mutates (kernel_update_head.py:125)
        mask_head=[copy.deepcopy(stage_cfg(C, F, heads, L, n_thing, n_stuff)) for _ in range(S)],
        train_cfg=None, test_cfg=test_cfg)
    head.init_weights()
    head.eval()
    return head


def build_kernel_head(ns, N_thing_q=100, C=256, n_thing=8, n_stuff=11, groups=32):
    L = n_thing + n_stuff
    head = ns.KernelHead(
        num_proposals=N_thing_q, num_classes=L, num_thing_classes=n_thing,
        num_stuff_classes=n_stuff, in_channels=C, out_channels=C, num_heads=8, num_cls_fcs=1,
        num_seg_convs=1, num_loc_convs=1, conv_kernel_size=1, with_depth=True,
        cat_stuff_mask=True, feat_downsample_stride=2, feat_refine_stride=1, feat_refine=False,
        use_binary=True, num_depth_convs=1, conv_normal_init=True, proposal_feats_with_obj=True,
        xavier_init_kernel=False, kernel_init_std=1, feat_transform_cfg=None,
        norm_cfg=dict(type="GN", num_groups=groups),
        loss_rank=dict(type="CrossEntropyLoss", use_sigmoid=False, loss_weight=0.1),
        loss_seg=dict(type="FocalLoss", use_sigmoid=True),
        loss_mask=dict(type="CrossEntropyLoss", use_sigmoid=True),
        loss_dice=dict(type="DiceLoss", loss_weight=4.0),
        loss_depth=dict(type="DepthLoss", loss_weight=5.0),
        localization_fpn=dict(type="IdentityNeck"), train_cfg=None, test_cfg=None)
    head.init_weights()
    head.eval()
    return head


def load_reference_neck():
    """SemanticFPNWrapper (polyphonic/funcs/semantic_fpn.py) + mmdet's SinePositionalEncoding
    (mmdet/models/utils/positional_encoding.py, vendored in the reference tree)."""
    ns = load_reference()
    for name in ("mmdet.models.utils",):
        if name not in sys.modules:
            _mod(name)

    def load(modname, relpath):
        spec = importlib.util.spec_from_file_location(modname, os.path.join(REF_ROOT, relpath))
        m = importlib.util.module_from_spec(spec)
        sys.modules[modname] = m
        parent, _, child = modname.rpartition(".")
        setattr(sys.modules[parent], child, m)
        spec.loader.exec_module(m)
        return m

    pe = load("mmdet.models.utils.positional_encoding", "mmdet/models/utils/positional_encoding.py")
    sf = load("polyphonic.funcs.semantic_fpn", "polyphonic/funcs/semantic_fpn.py")
    return types.SimpleNamespace(SemanticFPNWrapper=sf.SemanticFPNWrapper, SinePositionalEncoding=pe.SinePositionalEncoding)


def neck_cfg(C=256, groups=32, num_feats=128):
    """configs/_base_/models/polyphonic_former.py:78-96"""
    return dict(in_channels=C, feat_channels=C, out_channels=C, start_level=0, end_level=3, upsample_times=2,
                positional_encoding=dict(type="SinePositionalEncoding", num_feats=num_feats, normalize=True),
                cat_coors=False, cat_coors_level=3, fuse_by_cat=False, return_list=False, num_aux_convs=2,
                norm_cfg=dict(type="GN", num_groups=groups, requires_grad=True))


def load_reference_assigner():
    """polyphonic/funcs/assigner.py (MaskHungarianAssigner[WithDepth], DiceCost, MaskCost, DepthCost) together with the
    vendored mmdet FocalLossCost (mmdet/core/bbox/match_costs/match_cost.py:54-99), loaded by file path.  Stand-ins:
    `AssignResult` (a record of num_gts / gt_inds / max_overlaps / labels, mmdet/core/bbox/assigners/assign_result.py:41-48),
    `BaseAssigner` (an empty base), the MATCH_COST registry and `build_match_cost` (registry build)."""
    load_reference()
    core = sys.modules["mmdet.core"]

    class AssignResult:
        def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
            self.num_gts, self.gt_inds, self.max_overlaps, self.labels = num_gts, gt_inds, max_overlaps, labels
            self._extra = {}

        def set_extra_property(self, k, v):
            self._extra[k] = v

    core.AssignResult = AssignResult
    core.BaseAssigner = type("BaseAssigner", (), {})
    MATCH_COST = Registry("match_cost")
    mc = _mod("mmdet.core.bbox.match_costs")
    b = _mod("mmdet.core.bbox.match_costs.builder")
    b.MATCH_COST = MATCH_COST
    b.build_match_cost = lambda cfg, default_args=None: MATCH_COST.build(cfg)
    iou = _mod("mmdet.core.bbox.iou_calculators")
    iou.bbox_overlaps = None        # box costs are not on this path
    tr = _mod("mmdet.core.bbox.transforms")
    tr.bbox_cxcywh_to_xyxy = tr.bbox_xyxy_to_cxcywh = None

    def load(modname, relpath):
        spec = importlib.util.spec_from_file_location(modname, os.path.join(REF_ROOT, relpath))
        m = importlib.util.module_from_spec(spec)
        sys.modules[modname] = m
        parent, _, child = modname.rpartition(".")
        setattr(sys.modules[parent], child, m)
        spec.loader.exec_module(m)
        return m

    load("mmdet.core.bbox.match_costs.match_cost", "mmdet/core/bbox/match_costs/match_cost.py")
    a = load("polyphonic.funcs.assigner", "polyphonic/funcs/assigner.py")
    return types.SimpleNamespace(module=a, MATCH_COST=MATCH_COST, Assigner=a.MaskHungarianAssignerWithDepth,
                                 AssignerNoDepth=a.MaskHungarianAssigner, AssignResult=AssignResult)
