"""Goldens for SURVEY.md 8(f) N2 (wire format + DVPQ), produced by the UNMODIFIED reference evaluator
(/root/reference/datasets/utils.py:vpq_eval, polyphonic/apis/video_evaluate.py:video_evaluate) reading files that
THIS build's writer (polyphonicformer_amd.dvps_eval.save_record) produced.  Test infrastructure; runs only in the
build container (the reference is not available on the GPU box).

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_eval.py
"""
import contextlib, importlib.util, io, json, os, re, sys, tempfile, types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import helpers  # noqa: E402
from polyphonicformer_amd import dvps_eval as D  # noqa: E402


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def load_reference_evaluator():
    # stand-ins for what the evaluator imports but does not need for its arithmetic
    mmcv = types.ModuleType("mmcv")
    mmcv.scandir = lambda d: iter(sorted(os.listdir(d)))
    sys.modules.setdefault("mmcv", mmcv)
    for pkg in ("datasets", "polyphonic", "polyphonic.apis"):
        if pkg not in sys.modules:
            m = types.ModuleType(pkg)
            m.__path__ = []
            sys.modules[pkg] = m
    utils = _load("datasets.utils", os.path.join(REF, "datasets", "utils.py"))
    apis_utils = types.ModuleType("polyphonic.apis.utils")
    apis_utils.track_parallel_progress = lambda func, tasks, nproc, **k: [func(*t) for t in tasks]   # serial
    sys.modules["polyphonic.apis.utils"] = apis_utils
    ve = _load("polyphonic.apis.video_evaluate", os.path.join(REF, "polyphonic", "apis", "video_evaluate.py"))
    return utils, ve


def main():
    utils, ve = load_reference_evaluator()
    out = {}
    # 1) vpq_eval on single frames and on 2-frame clips
    frames = helpers.dvps_clip(seed=21)
    vp = {}
    for i, fr in enumerate(frames[:6]):
        p, g = D.wire_record(fr["pred"])["panseg"], D.wire_record(fr["gt"])["panseg"]
        r = utils.vpq_eval([p, g], num_classes=19)
        vp[f"frame{i}"] = np.stack(r)
    np.savez_compressed(os.path.join(REPO, "tests", "golden", "dvps_vpq.npz"), **vp)
    # 2) the whole evaluator on files written by this build's writer (torch >= 2.6 needs weights_only=False
    #    to read numpy arrays the way the reference's torch did)
    _tl = torch.load
    torch.load = lambda f, *a, **k: _tl(f, *a, **{**k, "weights_only": False})
    with tempfile.TemporaryDirectory() as d:
        for fr in frames:
            D.save_record(d, fr["seq"], fr["img"], fr["pred"], "pred")
            D.save_record(d, fr["seq"], fr["img"], fr["gt"], "gt")
        rec = torch.load(os.path.join(d, "pred", D.record_name(frames[0]["seq"], frames[0]["img"])))
        out["record_keys"] = sorted(rec.keys())
        out["record_dtypes"] = {k: str(v.dtype) for k, v in rec.items()}
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            ve.video_evaluate(d, ["DVPQ"], num_classes=19, num_things=8)
    torch.load = _tl
    lines = buf.getvalue().splitlines()
    res, cur = {}, None
    for ln in lines:
        m = re.match(r"Evaluating DVPQ: k=(\d+); lambda=(\S+)", ln)
        if m:
            cur = f"{m.group(1)}:{m.group(2)}"
        m = re.match(r"DVPQ : ([\d.]+) DVPQ_thing : ([\d.]+) DVPQ_stuff : ([\d.]+)", ln)
        if m and cur:
            res[cur] = [float(m.group(1)), float(m.group(2)), float(m.group(3))]
    out["dvpq"] = res
    # 3) depth error metrics
    g = np.concatenate([f["gt"]["depth"].ravel() for f in frames]); p = np.concatenate([f["pred"]["depth"].ravel() for f in frames])
    out["depth_errors"] = {k: float(v) for k, v in utils.compute_errors(p, g).items()}
    json.dump(out, open(os.path.join(REPO, "tests", "golden", "dvps_eval.json"), "w"), indent=1)
    print(json.dumps(out, indent=1)[:1500])


if __name__ == "__main__":
    main()
