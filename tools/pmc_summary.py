"""Summarises the separate --pmc FETCH_SIZE / WRITE_SIZE passes of tools/pool_only.py into pmc_traffic.json
(bytes per launch, gfx950 correction per MI355X_MICROARCH.md: hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024)."""
import csv, glob, json, os, sys
PART = int(os.environ.get("PH_PART_FRAMES", 32))
from collections import defaultdict
src, dst = sys.argv[1], sys.argv[2]
def per_kernel(path, counter):
    f = glob.glob(f"{src}/{path}/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    # round 6: tools/pool_only.py launches k_pool alternately over depth_feats alone (after k_dynconv_poolx) and over both maps -- the
    # same kernel name and, at the headline geometry, the same grid size (10 x 2 x 24 = 5 x 4 x 24 workgroups): told apart by launch order
    fused = any("k_dynconv_poolx" in r["Kernel_Name"] for r in rows)
    pool_ids = sorted({int(r["Dispatch_Id"]) for r in rows if "k_pool<" in r["Kernel_Name"]})
    depth_ids = set(pool_ids[0::2]) if fused else set()
    acc, n = defaultdict(float), defaultdict(set)
    for r in rows:
        k = r["Kernel_Name"]
        if int(r["Dispatch_Id"]) in depth_ids:
            k = "POOL_DEPTH " + k
        acc[k] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    return {k: acc[k] / len(n[k]) for k in acc}
fe, wr = per_kernel("pmc_fetch", "FETCH_SIZE"), per_kernel("pmc_write", "WRITE_SIZE")
import re
_conv = lambda k: re.search(r"k_dynconv<\d+, \d+, \d+, \d+, (true|false),", k)       # 5th template argument: BITS
_up2 = lambda k: re.search(r"k_dynconv_up2m?<\d+, \d+, \d+, (true|false),", k)         # 4th template argument: LOWRES (round 6: k_dynconv_up2m)
names = {"pool": lambda k: "k_pool" in k and not k.startswith("POOL_DEPTH"), "pool_depth": lambda k: k.startswith("POOL_DEPTH"),
         "dynconv_poolx": lambda k: "k_dynconv_poolx" in k, "dynconv_bits": lambda k: bool(_conv(k)) and _conv(k).group(1) == "true",
         "dynconv_logits": lambda k: bool(_conv(k)) and _conv(k).group(1) == "false",
         "upsample2x": lambda k: "k_upsample2x" in k,
         "dynconv_up2_mask": lambda k: bool(_up2(k)) and _up2(k).group(1) == "true",
         "dynconv_up2_depth": lambda k: bool(_up2(k)) and _up2(k).group(1) == "false"}
out = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes of `python tools/pool_only.py mixed16` (cfg2 shape, " + str(PART) + " frames per launch, the headline precision mode: bf16 feature planes, one fp16 plane of dynamic kernels, fp16 logits). Units are KiB; per MI355X_MICROARCH.md (HBM section) FETCH_SIZE reports 1/2 of the bytes of wide coalesced streaming reads on gfx950, so hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.",
       "frames_per_launch": PART, "mode": "mixed16", "kernels": {}}
for name, pred in names.items():
    f = [v for k, v in fe.items() if pred(k)]; w = [v for k, v in wr.items() if pred(k)]
    if f and w:
        out["kernels"][name] = {"FETCH_SIZE_KiB": f[0], "WRITE_SIZE_KiB": w[0], "hbm_bytes_per_launch": int((2 * f[0] + w[0]) * 1024)}
json.dump(out, open(dst, "w"), indent=1); print(json.dumps(out["kernels"], indent=1))
