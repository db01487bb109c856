#!/bin/bash
mkdir -p gpurun_out/px6
timeout 1500 python -m pytest tests -q -x -m gpu 2>&1 | tail -6 > gpurun_out/px6/tests.log; cat gpurun_out/px6/tests.log
Q="--no-cpu-baseline --no-kernel-head --no-neck --steps 30 --warmup 5"
for i in 1 2; do
  python bench.py $Q 2>/dev/null | tail -1 > gpurun_out/px6/ab_poolx_$i.json
  PH_CONV_POOLX=0 python bench.py $Q 2>/dev/null | tail -1 > gpurun_out/px6/ab_separate_$i.json
done
python - <<'P' | tee gpurun_out/px6/poolx_ab.txt
import json, glob
print("# same-box A/B, alternating: bench.py --no-cpu-baseline --no-kernel-head --no-neck --steps 30 --warmup 5 (cfg2, mixed16, 96 frames per step)")
for f in sorted(glob.glob("gpurun_out/px6/ab_*.json")):
    d = json.load(open(f)); print(f.split("/")[-1], d["value"], "frames/s", d["ms_per_step"], "ms per step", {k: d["kernels_ms"][k] for k in d["kernels_ms"] if "pool" in k or "dynconv_bits" in k})
P
python tools/px_time.py | tee gpurun_out/px6/px_time.txt
