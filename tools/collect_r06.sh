#!/bin/bash
# Round 6 evidence, one gpurun call, one box -> gpurun_out/r06/ (copy what should be judged into profiles/r06/).
#   bash tools/collect_r06.sh
set -u
OUT=gpurun_out/r06; mkdir -p $OUT
export TMPDIR=/tmp
Q="--no-cpu-baseline --no-kernel-head --no-neck --steps 30 --warmup 5"
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
# same-box A/B of the fused final stage: round 4's window-pass kernel against the second-MFMA-product kernel (alternating)
for rep in 1 2; do
  python bench.py $Q > $OUT/ab_up2m_$rep.json 2>/dev/null
  PH_UP2_MFMA=0 python bench.py $Q > $OUT/ab_window_$rep.json 2>/dev/null
done
python - <<'PY' > $OUT/ab_final_stage.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r06/ab_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], d["value"], "frames/s", d["ms_per_step"], "ms per 96 frames", {k: d["kernels_ms"][k] for k in ("dynconv_up2_mask", "dynconv_up2_depth")})
PY
cat $OUT/ab_final_stage.txt
# the overlap timeline of the 4-stream graph + the single-stream trace of one part
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/d -o d -- python bench.py $Q > $OUT/d_bench.json 2> $OUT/d.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/e -o e -- python bench.py $Q --streams 1 --frames 24 > $OUT/e_bench.json 2> $OUT/e.err
python tools/timeline.py $(find $OUT/d -name "*kernel_trace.csv") --isolated $(find $OUT/e -name "*kernel_trace.csv") --json $OUT/timeline_4streams.json > /dev/null
# cfg5: 4 streams + single stream, its timeline
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5 -o c5 -- python bench.py $Q --workload cfg5 --precision fp16 --frames 192 > $OUT/c5_bench.json 2> $OUT/c5.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5s -o c5s -- python bench.py $Q --workload cfg5 --precision fp16 --frames 48 --streams 1 > $OUT/c5s_bench.json 2> $OUT/c5s.err
python tools/timeline.py $(find $OUT/c5 -name "*kernel_trace.csv") --isolated $(find $OUT/c5s -name "*kernel_trace.csv") --json $OUT/timeline_cfg5.json > /dev/null
# HBM traffic + SQ counters of the a6 kernels at the headline geometry (separate passes, counters only)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o p -- python tools/pool_only.py mixed16 > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o p -- python tools/pool_only.py mixed16 > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_sq -o p -- python tools/pool_only.py mixed16 > /dev/null 2> $OUT/pmc_sq.err
python tools/pmc_sq_summary.py $OUT/pmc_sq > $OUT/pmc_sq_summary.txt 2>&1
python tools/pmc_summary.py $OUT $OUT/pmc_traffic.json > $OUT/pmc_summary.txt 2>&1
python tools/r04_kernels.py mixed16 > $OUT/kernels_isolated.json 2> $OUT/kernels_isolated.err; PH_UP2_MFMA=0 python tools/r04_kernels.py mixed16 >> $OUT/kernels_isolated.json 2>> $OUT/kernels_isolated.err
# video
python tools/clip_phases.py 8 fp16 > $OUT/clip_phases.txt 2> $OUT/clip_phases.err; python tools/clip_phases.py 2 fp16 >> $OUT/clip_phases.txt 2>> $OUT/clip_phases.err
python bench.py --workload cfg4 --steps 40 --warmup 4 > $OUT/bench_cfg4_world1.json 2> $OUT/bench_cfg4.err
python bench.py --workload cfg4 --steps 40 --warmup 4 --clip-frames 8 > $OUT/bench_cfg4_world1_clip8.json 2> $OUT/bench_cfg4_clip8.err
PH_CFG4_EARLY_BEGIN=1 python bench.py --workload cfg4 --steps 40 --warmup 4 --clip-frames 8 --no-cpu-baseline > $OUT/bench_cfg4_clip8_early_begin.json 2>/dev/null
PH_CFG4_EARLY_BEGIN=1 PH_VIDEO_SLOT_PRIO=low python bench.py --workload cfg4 --steps 40 --warmup 4 --clip-frames 8 --no-cpu-baseline > $OUT/bench_cfg4_clip8_early_begin_low_prio.json 2>/dev/null
PH_VIDEO_SLOT_PRIO=low python bench.py --workload cfg4 --steps 40 --warmup 4 --clip-frames 8 --no-cpu-baseline > $OUT/bench_cfg4_clip8_low_prio.json 2>/dev/null
python - <<'PY' > $OUT/cfg4_orders.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r06/bench_cfg4*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], d["value"], "frames/s", d["ms_per_step"], "ms per step")
    except Exception as e:
        print(f, "failed", e)
PY
cat $OUT/cfg4_orders.txt
# the conv's ring: one workgroup that owns the CU (4-deep ring) against two half-CU workgroups (2-deep)
for rep in 1 2; do
  python bench.py $Q 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('conv ring 4, 256 wgs', d['value'], d['kernels_ms']['dynconv_bits'])"
  PH_ALT_LIB=tools/libpolyhead_conv2.so PH_CONV_WGS=512 python bench.py $Q 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('conv ring 2, 512 wgs', d['value'], d['kernels_ms']['dynconv_bits'])"
done > $OUT/conv_ring_ab.txt 2>&1
cat $OUT/conv_ring_ab.txt
if [ "${ALL_LEGS:-1}" = 1 ]; then python bench.py --all-legs > $OUT/bench_all_legs.json 2> $OUT/bench_all_legs.err; fi
find $OUT -name "*.csv" -size +20M -delete
ls $OUT | head -80
