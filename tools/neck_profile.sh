#!/bin/bash
# per-kernel time of one SemanticFPNWrapper.forward (B = 8, cfg2 sizes) from a rocprofv3 kernel trace
export TMPDIR=/tmp
rm -rf gpurun_out/nk
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/nk -o nk -- python bench.py --no-cpu-baseline --no-kernel-head --steps 3 --warmup 1 > gpurun_out/nk_bench.json 2>/dev/null
python - <<PY
import csv, json
tot = 0
rows = list(csv.DictReader(open("gpurun_out/nk/nk_kernel_stats.csv")))
nfwd = [int(r["Calls"]) for r in rows if "k_gn_sum_planes" in r["Name"]][0]      # one launch per forward
for r in rows:
    if "nhwc" in r["Name"] or "gn_" in r["Name"]:
        us = int(r["TotalDurationNs"]) // (1000 * nfwd)
        tot += us
        print(r["Name"][:58].ljust(58), r["Calls"].rjust(4), str(round(float(r["AverageNs"]) / 1e3, 1)).rjust(8), str(us).rjust(6), "us/forward")
print("sum", tot, "us/forward;", json.loads(open("gpurun_out/nk_bench.json").read().strip().split("\n")[-1]).get("semantic_fpn_neck"))
PY
