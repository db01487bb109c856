#!/usr/bin/env python
"""TEST INFRASTRUCTURE ONLY -- dumps the `model` dictionaries of the reference's OWN config files as data.

Runs (runpy, in this container) the reference's shipped configs

    configs/polyphonic_image/poly_r50_cityscapes_2x.py     (BASELINE configs[0] / [1]: the image head)
    configs/polyphonic_video/poly_r50_cityscapes_1x.py     (configs[2] / [3]: the video head, tracker, track head)

resolving `_base_` the way mmcv.Config.fromfile does (bases loaded first, the child's dictionaries merged INTO the base's key by
key, `_delete_=True` replaces), and writes the resulting `model` dict of each -- kwargs only, no source text -- to
`tests/golden/ref_model_cfg.json`.  `tests/test_ref_configs.py` builds every head of this package from that JSON exactly as
`TwoStageDetector.__init__` does (mmdet/models/detectors/two_stage.py:36-49: rpn_head.update(train_cfg=train_cfg.rpn,
test_cfg=test_cfg.rpn), roi_head.update(train_cfg=train_cfg.rcnn, test_cfg=test_cfg.rcnn)) and `PolyphonicVideo.__init__`
(polyphonic/polyphonic_former_video.py:49-60), so the kwargs contract of the boundary is pinned by the reference's files and not
by hand-written dictionaries.

usage: python oracle/gen_ref_cfg.py        (needs /root/reference; the JSON it writes is committed)"""
import json
import os
import runpy
import sys

REF = os.environ.get("POLY_REFERENCE_ROOT", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ref_model_cfg.json")


def _merge(child, base):
    """mmcv.Config._merge_a_into_b: dictionaries merge recursively, `_delete_` replaces, everything else overwrites"""
    out = dict(base)
    for k, v in child.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict) and not v.get("_delete_", False):
            out[k] = _merge(v, out[k])
        else:
            out[k] = {kk: vv for kk, vv in v.items() if kk != "_delete_"} if isinstance(v, dict) else v
    return out


def load_cfg(path):
    ns = runpy.run_path(path)
    cfg = {k: v for k, v in ns.items() if not k.startswith("__") and isinstance(v, (dict, list, tuple, str, int, float, bool, type(None)))}
    bases = cfg.pop("_base_", [])
    if isinstance(bases, str):
        bases = [bases]
    base = {}
    for b in bases:
        bc = load_cfg(os.path.normpath(os.path.join(os.path.dirname(path), b)))
        dup = set(base) & set(bc)
        if dup:
            raise KeyError(f"duplicate keys in the bases of {path}: {sorted(dup)}")
        base.update(bc)
    return _merge(cfg, base)


def jsonable(v):
    if isinstance(v, dict):
        return {k: jsonable(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [jsonable(x) for x in v]
    return v


def main():
    files = {"image": "configs/polyphonic_image/poly_r50_cityscapes_2x.py", "video": "configs/polyphonic_video/poly_r50_cityscapes_1x.py"}
    out = {"_source": {k: v for k, v in files.items()},
           "_note": "model dicts of the reference's shipped configs after _base_ resolution (oracle/gen_ref_cfg.py); data only"}
    for name, rel in files.items():
        model = load_cfg(os.path.join(REF, rel))["model"]
        out[name] = jsonable({k: v for k, v in model.items() if k not in ("backbone", "neck")})      # backbone / FPN: out of scope
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")
    print(OUT, {k: sorted(v) for k, v in out.items() if not k.startswith("_")})


if __name__ == "__main__":
    sys.exit(main())
