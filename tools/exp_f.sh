#!/bin/bash
# round 6, GPU call F: k_dynconv_up2m variants + timing ablations, cfg4 with queued clips, the whole GPU suite
set -u
OUT=gpurun_out/r06f; mkdir -p $OUT
export TMPDIR=/tmp
Q="--no-cpu-baseline --no-kernel-head --no-neck --steps 30 --warmup 5"
val() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', d['value'], d['ms_per_step'], {k: round(v,4) for k,v in d['kernels_ms'].items()})" 2>&1 | tail -1; }
k4() { python tools/r04_kernels.py mixed16 2>> $OUT/k.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['up2_mask_us'], d['up2_depth_us'])"; }
for rep in 1 2; do
  python bench.py $Q > $OUT/b_il$rep.json 2> $OUT/b_il$rep.err; val $OUT/b_il$rep.json interleaved$rep
  PH_ALT_LIB=tools/libpolyhead_upmser.so python bench.py $Q > $OUT/b_ser$rep.json 2> $OUT/b_ser$rep.err; val $OUT/b_ser$rep.json serial$rep
done
k4 interleaved; PH_ALT_LIB=tools/libpolyhead_upmser.so k4 serial
export PH_ALT_LIB=tools/libpolyhead_upmtime.so
k4 timing_lib_full; PH_UP2_DBG=1 k4 no_stores; PH_UP2_DBG=2 k4 no_emission; PH_UP2_DBG=4 k4 all_silent
unset PH_ALT_LIB
python bench.py --workload cfg4 --steps 40 --warmup 4 --clip-frames 8 --no-cpu-baseline > $OUT/bench_cfg4_clip8.json 2> $OUT/bench_cfg4_clip8.err
python -c "
import json; d=json.loads(open('$OUT/bench_cfg4_clip8.json').read().strip().splitlines()[-1]); print('cfg4 clip8', d['value'], d['ms_per_step'], {k: d['cfg4'][k] for k in ('heads_merge_records_ms_per_step','replay_tracking_ms_per_step','allgather_track_records_us_per_step','khead_onepass_timeouts')})"
PH_CFG4_LATE_BEGIN=1 python bench.py --workload cfg4 --steps 40 --warmup 4 --clip-frames 8 --no-cpu-baseline > $OUT/bench_cfg4_clip8_late.json 2> $OUT/bench_cfg4_clip8_late.err
python -c "
import json; d=json.loads(open('$OUT/bench_cfg4_clip8_late.json').read().strip().splitlines()[-1]); print('cfg4 clip8 late-begin', d['value'], d['ms_per_step'])"
python bench.py --workload cfg4 --steps 40 --warmup 4 --clip-frames 2 --no-cpu-baseline > $OUT/bench_cfg4_clip2.json 2> $OUT/bench_cfg4_clip2.err
python -c "
import json; d=json.loads(open('$OUT/bench_cfg4_clip2.json').read().strip().splitlines()[-1]); print('cfg4 clip2', d['value'], d['ms_per_step'])"
timeout 1500 python -m pytest tests -m gpu -q --durations=12 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -22 $OUT/pytest_gpu.log
