#!/bin/bash
# round 6, GPU call D: fused-final-stage helper variants (same-box A/B), neck after the pair-wise partial sums, dist-test timing, cfg5 sweep
set -u
OUT=gpurun_out/r06d; mkdir -p $OUT
export TMPDIR=/tmp
Q="--no-cpu-baseline --no-kernel-head --no-neck --steps 30 --warmup 5"
val() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', d['value'], d['ms_per_step'], {k: round(v,4) for k,v in d['kernels_ms'].items()})" 2>&1 | tail -1; }
for rep in 1 2; do
  for v in base balp bals nobal; do
    if [ $v = base ]; then unset PH_ALT_LIB; else export PH_ALT_LIB=tools/libpolyhead_$v.so; fi
    python bench.py $Q > $OUT/b_$v$rep.json 2> $OUT/b_$v$rep.err; val $OUT/b_$v$rep.json $v$rep
  done
done
for v in base balp bals nobal; do
  if [ $v = base ]; then unset PH_ALT_LIB; else export PH_ALT_LIB=tools/libpolyhead_$v.so; fi
  python tools/r04_kernels.py mixed16 2> $OUT/k_$v.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['up2_mask_us'], d['up2_depth_us'])"
done
unset PH_ALT_LIB
timeout 900 python -m pytest tests/test_gpu_neck.py tests/test_gpu_neck_train.py tests/test_gpu_dist.py tests/test_gpu_kernels.py -q --durations=8 -k "not two_rank" > $OUT/pytest_a.log 2>&1; echo "pytest a rc $?"; tail -14 $OUT/pytest_a.log
timeout 900 python -m pytest tests/test_gpu_video.py tests/test_gpu_parity.py -q --durations=5 > $OUT/pytest_b.log 2>&1; echo "pytest b rc $?"; tail -10 $OUT/pytest_b.log
python tools/fullhead_leg.py > $OUT/fullhead.txt 2>&1; grep "neck\|full" $OUT/fullhead.txt | cut -c1-200
for fr in 192 256 384; do
  python bench.py $Q --workload cfg5 --precision fp16 --frames $fr > $OUT/c5_$fr.json 2> $OUT/c5_$fr.err; val $OUT/c5_$fr.json cfg5_f$fr
done
PH_QUERY_NRT=2 python bench.py $Q --workload cfg5 --precision fp16 --frames 256 > $OUT/c5_nrt2.json 2> $OUT/c5_nrt2.err; val $OUT/c5_nrt2.json cfg5_f256_nrt2
PH_QUERY_NRT=8 python bench.py $Q --workload cfg5 --precision fp16 --frames 256 > $OUT/c5_nrt8.json 2> $OUT/c5_nrt8.err; val $OUT/c5_nrt8.json cfg5_f256_nrt8
python bench.py $Q --workload cfg5 --precision fp16 --frames 256 --streams 2 > $OUT/c5_s2.json 2> $OUT/c5_s2.err; val $OUT/c5_s2.json cfg5_f256_s2
python bench.py $Q --workload cfg5 --precision fp16 --frames 256 --streams 8 > $OUT/c5_s8.json 2> $OUT/c5_s8.err; val $OUT/c5_s8.json cfg5_f256_s8
