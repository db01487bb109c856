#!/bin/bash
# round 6, GPU call B: the whole GPU suite with this round's new tests, the default bench line, cfg5 under the profiler
set -u
OUT=gpurun_out/r06b; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -5 $OUT/pytest_gpu.log
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc $?"
python -c "
import json; d=json.loads(open('$OUT/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']); print(d.get('cfg5_fp16')); print(d.get('sparse_masks'))"
Q="--no-cpu-baseline --no-kernel-head --no-neck --steps 20 --warmup 5"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5 -o c5 -- python bench.py $Q --workload cfg5 --precision fp16 --frames 192 > $OUT/c5_bench.json 2> $OUT/c5.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5s -o c5s -- python bench.py $Q --workload cfg5 --precision fp16 --frames 48 --streams 1 > $OUT/c5s_bench.json 2> $OUT/c5s.err
python tools/timeline.py $(find $OUT/c5 -name "*kernel_trace.csv") --isolated $(find $OUT/c5s -name "*kernel_trace.csv") --json $OUT/timeline_cfg5.json > /dev/null
head -12 $(find $OUT/c5s -name "*kernel_stats.csv") | cut -c1-160
find $OUT -name "*.csv" -size +20M -delete
