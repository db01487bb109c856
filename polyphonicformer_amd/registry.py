"""Registry surface of the drop-in.  The reference registers its heads in mmdet's
shared `MODELS` registry (mmdet/models/builder.py:7-16) and `KernelUpdator` in mmcv's
`TRANSFORMER_LAYER` (polyphonic/funcs/kernel_updator.py:6).  When mmcv/mmdet are importable the
classes of this package are registered there under the SAME names (`force=True`, so that they
replace the Python originals); a build-owned registry with the same `register_module` / `build`
API always exists, because neither box of this build has mmcv."""
import copy


class ConfigDict(dict):
    """dict with (nested) attribute access, the part of mmcv.ConfigDict the heads rely on
    (`test_cfg.max_per_img`, `test_cfg.merge_stuff_thing.overlap_thr`, kernel_update.py:431,503-511)."""

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
        def _reg(cls):
            key = name or cls.__name__
            if key in self._module_dict and not force:
                raise KeyError(f"{key} is already registered in {self.name}")
            self._module_dict[key] = cls
            return cls

        if module is not None:
            return _reg(module)
        return _reg

    def get(self, key):
        return self._module_dict.get(key)

    def __contains__(self, key):
        return key in self._module_dict

    def build(self, cfg, default_args=None):
        if not isinstance(cfg, dict) or "type" not in cfg:
            raise TypeError(f"cfg must be a dict with a `type` key, got {cfg!r}")
        cfg = dict(cfg)
        if default_args:
            for k, v in default_args.items():
                cfg.setdefault(k, v)
        typ = cfg.pop("type")
        cls = self._module_dict.get(typ) if isinstance(typ, str) else typ
        if cls is None:
            raise KeyError(f"{typ} is not in the {self.name} registry")
        return cls(**cfg)


MODELS = Registry("models")
HEADS = NECKS = LOSSES = BACKBONES = DETECTORS = MODELS        # aliases, as in mmdet
TRANSFORMER_LAYER = Registry("transformer_layer")


def build_head(cfg):
    return HEADS.build(cfg)


def build_neck(cfg):
    return NECKS.build(cfg)


def build_loss(cfg):
    return LOSSES.build(cfg)


def build_transformer_layer(cfg, default_args=None):
    return TRANSFORMER_LAYER.build(cfg, default_args)


def register_everywhere(cls, kind="head"):
    """register under the reference's name here and, if present, in mmdet / mmcv"""
    (TRANSFORMER_LAYER if kind == "transformer_layer" else MODELS).register_module(module=cls, force=True)
    try:  # pragma: no cover - mmcv/mmdet are not installed on the build or GPU boxes
        if kind == "transformer_layer":
            from mmcv.cnn.bricks.transformer import TRANSFORMER_LAYER as T
            T.register_module(module=cls, force=True)
        else:
            from mmdet.models.builder import HEADS as Hd
            Hd.register_module(module=cls, force=True)
    except Exception:
        pass
    return cls


def deep_cfg(cfg):
    """configs are mutated by the heads (feat_transform_cfg.pop, kernel_update_head.py:125): copy first"""
    return copy.deepcopy(cfg)


# ---- the reference's mixed-precision hook ---------------------------------------------------------------------------------
def wrap_fp16_model(model):
    """mmcv.runner.wrap_fp16_model as the reference applies it (tools/test.py:202-204, tools/test_video.py: `fp16_cfg =
    cfg.get('fp16', None); if fp16_cfg is not None: wrap_fp16_model(model)`): every module of this package under `model`
    that has precision grades is switched to its fp16 grade -- KernelHead (and its neck) to one fp16 plane of maps and
    weights, KernelUpdateIterHead to the `fp16` mode of engine.MODES with fp16 logits, the track head to the grade engine.PREC maps 'fp16' to.
    `model`: a module tree (a detector holding `rpn_head` / `roi_head`, or a head), or an iterable of modules."""
    import torch
    mods = list(model) if isinstance(model, (list, tuple)) else [model]
    done = []
    for root in mods:
        for m in root.modules():
            if m.__class__.__name__ == "KernelUpdateIterHead" and hasattr(m, "set_precision"):
                m.set_precision("fp16", torch.float16)
                done.append(m)
            elif m.__class__.__name__ == "KernelHead" and hasattr(m, "set_precision"):
                m.set_precision("fp16")          # forwards to its SemanticFPNWrapper
                done.append(m)
            elif m.__class__.__name__ == "QuasiDenseMaskEmbedHeadGTMask" and hasattr(m, "precision"):
                m.precision = "fp16"             # engine.PREC: the split grade of the track head (its embeddings feed a hard match)
                done.append(m)
    return done


def build_heads_from_config(cfg):
    """The two heads of a reference detector config (`cfg.model.rpn_head`, `cfg.model.roi_head`, with `train_cfg` /
    `test_cfg` injected as TwoStageDetector does, mmdet/models/detectors/two_stage.py:36-49) built from this package's
    registry, honouring the config's `fp16` key exactly as tools/test.py:202-204 does.  `cfg`: the loaded config as a
    (nested) dict.  Returns (rpn_head, roi_head)."""
    # the modules whose import registers the names a reference config uses (the reference's `polyphonic/__init__.py` does the same)
    from . import kernel_head, kernel_update, kernel_update_head, kernel_updator, semantic_fpn, assigner, losses  # noqa: F401
    model = cfg["model"]
    train_cfg, test_cfg = model.get("train_cfg"), model.get("test_cfg")
    rpn = deep_cfg(model["rpn_head"])
    roi = deep_cfg(model["roi_head"])
    if train_cfg is not None:
        rpn.setdefault("train_cfg", deep_cfg(train_cfg.get("rpn")))
        roi.setdefault("train_cfg", deep_cfg(train_cfg.get("rcnn")))
    if test_cfg is not None:
        rpn.setdefault("test_cfg", deep_cfg(test_cfg.get("rpn")))
        roi.setdefault("test_cfg", deep_cfg(test_cfg.get("rcnn")))
    rpn_head, roi_head = build_head(rpn), build_head(roi)
    if cfg.get("fp16", None) is not None:
        wrap_fp16_model([rpn_head, roi_head])
    return rpn_head, roi_head
