#!/bin/bash
# round 6: fused conv + pooling-of-x kernel -- parity tests, then same-box A/B of the step
mkdir -p gpurun_out/px
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "poolx" 2>&1 | tail -15 > gpurun_out/px/test.log
cat gpurun_out/px/test.log
for i in 1 2; do
for v in 0 1; do
  PH_CONV_POOLX=$v timeout 300 python bench.py --no-cpu-baseline --no-kernel-head --no-neck --steps 30 --warmup 5 2>gpurun_out/px/err_$v.log | tail -1 > gpurun_out/px/bench_${v}_$i.json
  python - <<P
import json
try:
    d=json.load(open("gpurun_out/px/bench_${v}_$i.json")); print("POOLX=$v run $i", d["value"], d["ms_per_step"])
except Exception as e: print("fail $v", e); print(open("gpurun_out/px/err_$v.log").read()[-2000:])
P
done; done
python - <<P
import json
for v in (0,1):
    d=json.load(open(f"gpurun_out/px/bench_{v}_2.json"))
    print(v, json.dumps(d.get("kernel_ms_per_launch", {k: d[k] for k in d if "kernel" in k}))[:2500])
P
