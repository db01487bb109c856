#!/bin/bash
# Round 6 evidence AFTER the fused conv + pooling kernel (k_dynconv_poolx) went into the step: one gpurun call, one box -> gpurun_out/r06b/
#   bash tools/collect_r06b.sh          (tools/collect_r06.sh collected the state before it: profiles/r06/*_before_poolx*)
set -u
OUT=gpurun_out/r06b; mkdir -p $OUT
export TMPDIR=/tmp
Q="--no-cpu-baseline --no-kernel-head --no-neck --steps 30 --warmup 5"
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/d -o d -- python bench.py $Q > $OUT/d_bench.json 2> $OUT/d.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/e -o e -- python bench.py $Q --streams 1 --frames 32 > $OUT/e_bench.json 2> $OUT/e.err
python tools/timeline.py $(find $OUT/d -name "*kernel_trace.csv") --isolated $(find $OUT/e -name "*kernel_trace.csv") --json $OUT/timeline_4streams.json > /dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o p -- python tools/pool_only.py mixed16 > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o p -- python tools/pool_only.py mixed16 > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_sq -o p -- python tools/pool_only.py mixed16 > /dev/null 2> $OUT/pmc_sq.err
python tools/pmc_sq_summary.py $OUT/pmc_sq > $OUT/pmc_sq_summary.txt 2>&1
python tools/pmc_summary.py $OUT $OUT/pmc_traffic.json > $OUT/pmc_summary.txt 2>&1
python tools/r04_kernels.py mixed16 > $OUT/kernels_isolated.json 2> $OUT/kernels_isolated.err
python bench.py --workload cfg4 --steps 40 --warmup 4 > $OUT/bench_cfg4_world1.json 2> $OUT/bench_cfg4.err
python bench.py --workload cfg4 --steps 40 --warmup 4 --clip-frames 8 --no-cpu-baseline > $OUT/bench_cfg4_world1_clip8.json 2> $OUT/bench_cfg4_clip8.err
python bench.py --workload cfg4 --steps 30 --warmup 4 --clip-frames 16 --no-cpu-baseline > $OUT/bench_cfg4_world1_clip16.json 2> $OUT/bench_cfg4_clip16.err
if [ "${ALL_LEGS:-1}" = 1 ]; then python bench.py --all-legs > $OUT/bench_all_legs.json 2> $OUT/bench_all_legs.err; fi
find $OUT -name "*.csv" -size +20M -delete
python - <<'PY'
import json
for f in ("bench_default", "bench_cfg4_world1", "bench_cfg4_world1_clip8", "bench_cfg4_world1_clip16", "bench_all_legs"):
    try:
        d = json.loads(open(f"gpurun_out/r06b/{f}.json").read().strip().splitlines()[-1]); print(f, d["value"], d["unit"], d["ms_per_step"], d.get("roofline"))
    except Exception as e:
        print(f, "failed", e)
print(open("gpurun_out/r06b/timeline_4streams.json").read()[:1500])
print(open("gpurun_out/r06b/pmc_summary.txt").read()[-1500:])
PY
