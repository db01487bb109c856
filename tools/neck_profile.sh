#!/bin/bash
# per-kernel time of one SemanticFPNWrapper.forward (cfg2 sizes) from a rocprofv3 kernel trace.  usage: bash tools/neck_profile.sh [precision] [B]
export TMPDIR=/tmp
R=$PWD; P=${1:-fp16}; B=${2:-16}
rm -rf /tmp/nk; mkdir -p gpurun_out
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/nk -o nk -- python $R/tools/neck_only.py $P $B > /tmp/nk_bench.json 2>/dev/null
cd $R
python - <<PY
import csv, json, glob
rows = list(csv.DictReader(open(glob.glob("/tmp/nk/**/nk_kernel_stats.csv", recursive=True)[0])))
nfwd = [int(r["Calls"]) for r in rows if "k_gn_sum_planes" in r["Name"] or "k_gn_sum_cplanes" in r["Name"]][0]      # one launch per forward
tot = 0
for r in rows:
    if "nhwc" in r["Name"] or "gn_" in r["Name"] or "k_khead" in r["Name"]:
        us = int(r["TotalDurationNs"]) // (1000 * nfwd)
        tot += us
        print(r["Name"][:70].ljust(70), str(int(r["Calls"]) // nfwd).rjust(3), "x", str(round(float(r["AverageNs"]) / 1e3, 1)).rjust(8), str(us).rjust(6), "us/forward")
print("sum", tot, "us/forward ($P, B = $B);", open("/tmp/nk_bench.json").read().strip().split("\n")[-1][:160])
PY
