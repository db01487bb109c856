#!/bin/bash
# round 6, GPU call C: balanced fused final stage (tests + same-box A/B), batch-invariant heads, cfg5 profile, neck / cfg4 after the tile changes
set -u
OUT=gpurun_out/r06c; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "up2" > $OUT/pytest_up2.log 2>&1; echo "up2 tests rc $?"; tail -3 $OUT/pytest_up2.log
Q="--no-cpu-baseline --no-kernel-head --no-neck --steps 30 --warmup 5"
val() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', d['value'], d['ms_per_step'], {k: round(v,4) for k,v in d['kernels_ms'].items()})" 2>&1 | tail -1; }
for rep in 1 2; do
  python bench.py $Q > $OUT/b_bal$rep.json 2> $OUT/b_bal$rep.err; val $OUT/b_bal$rep.json bal$rep
  PH_ALT_LIB=tools/libpolyhead_nobal.so python bench.py $Q > $OUT/b_nobal$rep.json 2> $OUT/b_nobal$rep.err; val $OUT/b_nobal$rep.json nobal$rep
done
python tools/r04_kernels.py mixed16 > $OUT/k_bal.json 2> $OUT/k_bal.err; echo bal; tail -1 $OUT/k_bal.json
PH_ALT_LIB=tools/libpolyhead_nobal.so python tools/r04_kernels.py mixed16 > $OUT/k_nobal.json 2> $OUT/k_nobal.err; echo nobal; tail -1 $OUT/k_nobal.json
timeout 1500 python -m pytest tests -m gpu -q --durations=25 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -45 $OUT/pytest_gpu.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5s -o c5s -- python bench.py $Q --workload cfg5 --precision fp16 --frames 48 --streams 1 > $OUT/c5s_bench.json 2> $OUT/c5s.err
head -12 $(find $OUT/c5s -name "*kernel_stats.csv") | cut -c1-150
python bench.py $Q --workload cfg5 --precision fp16 --frames 192 > $OUT/c5_bench.json 2> $OUT/c5.err; val $OUT/c5_bench.json cfg5
python tools/fullhead_leg.py > $OUT/fullhead.txt 2>&1; cat $OUT/fullhead.txt | cut -c1-250
python bench.py --workload cfg4 --steps 40 --warmup 4 --clip-frames 8 --no-cpu-baseline > $OUT/bench_cfg4_clip8.json 2> $OUT/bench_cfg4_clip8.err
python -c "
import json; d=json.loads(open('$OUT/bench_cfg4_clip8.json').read().strip().splitlines()[-1]); print('cfg4 clip8', d['value'], d['ms_per_step'], d['cfg4'])"
PH_VIDEO_CLIP_BATCH=3 python bench.py --workload cfg4 --steps 40 --warmup 4 --clip-frames 8 --no-cpu-baseline > $OUT/bench_cfg4_clip8_cap3.json 2> $OUT/bench_cfg4_clip8_cap3.err
python -c "
import json; d=json.loads(open('$OUT/bench_cfg4_clip8_cap3.json').read().strip().splitlines()[-1]); print('cfg4 clip8 cap3', d['value'], d['ms_per_step'])"
find $OUT -name "*.csv" -size +20M -delete
