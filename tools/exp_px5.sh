#!/bin/bash
mkdir -p gpurun_out/px5
python tools/px_time.py
Q="--no-cpu-baseline --no-kernel-head --no-neck --steps 30 --warmup 5"
for n in 10 16 21 32; do
  PH_POOLX_NSPLIT=$n python bench.py $Q 2>/dev/null | tail -1 > gpurun_out/px5/ns_$n.json
done
python - <<'P'
import json, glob
for f in sorted(glob.glob("gpurun_out/px5/*.json")):
    d = json.load(open(f)); print(f.split("/")[-1], d["value"], d["ms_per_step"], {k: d["kernels_ms"][k] for k in d["kernels_ms"] if "pool" in k or "query" in k})
P
