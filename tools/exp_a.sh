#!/bin/bash
# round 6, GPU call A: knob sweep of the 4-stream step + which SIMD bounds the fused final stage + the overlap timeline
set -u
OUT=gpurun_out/r06a; mkdir -p $OUT
export TMPDIR=/tmp
Q="--no-cpu-baseline --no-kernel-head --no-neck --steps 30 --warmup 5"
val() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', d['value'], d['ms_per_step'], {k: round(v,4) for k,v in d['kernels_ms'].items()})" 2>&1 | tail -1; }
python bench.py $Q > $OUT/b0.json 2> $OUT/b0.err; val $OUT/b0.json base
PH_QUERY_NRT=2 python bench.py $Q > $OUT/b_nrt2.json 2> $OUT/b_nrt2.err; val $OUT/b_nrt2.json nrt2
PH_QUERY_NRT=1 python bench.py $Q > $OUT/b_nrt1.json 2> $OUT/b_nrt1.err; val $OUT/b_nrt1.json nrt1
python bench.py $Q --streams 6 > $OUT/b_s6.json 2> $OUT/b_s6.err; val $OUT/b_s6.json s6_96
python bench.py $Q --streams 8 > $OUT/b_s8.json 2> $OUT/b_s8.err; val $OUT/b_s8.json s8_96
python bench.py $Q --streams 3 > $OUT/b_s3.json 2> $OUT/b_s3.err; val $OUT/b_s3.json s3_96
python bench.py $Q --frames 128 > $OUT/b_f128.json 2> $OUT/b_f128.err; val $OUT/b_f128.json s4_128
python bench.py $Q --frames 144 --streams 6 > $OUT/b_f144.json 2> $OUT/b_f144.err; val $OUT/b_f144.json s6_144
python bench.py $Q > $OUT/b1.json 2> $OUT/b1.err; val $OUT/b1.json base_again
for v in skiprt4 skiprt3; do
  PH_ALT_LIB=tools/libpolyhead_$v.so python tools/r04_kernels.py mixed16 > $OUT/k_$v.json 2> $OUT/k_$v.err; echo $v; tail -1 $OUT/k_$v.json
  PH_ALT_LIB=tools/libpolyhead_$v.so python bench.py $Q > $OUT/b_$v.json 2> $OUT/b_$v.err; val $OUT/b_$v.json $v
done
python tools/r04_kernels.py mixed16 > $OUT/k_base.json 2> $OUT/k_base.err; echo base; tail -1 $OUT/k_base.json
rocprofv3 --kernel-trace --output-format csv -d $OUT/d -o d -- python bench.py $Q > $OUT/d_bench.json 2> $OUT/d.err
rocprofv3 --kernel-trace --output-format csv -d $OUT/e -o e -- python bench.py $Q --streams 1 --frames 24 > $OUT/e_bench.json 2> $OUT/e.err
python tools/timeline.py $(find $OUT/d -name "*kernel_trace.csv") --isolated $(find $OUT/e -name "*kernel_trace.csv") --json $OUT/timeline.json | head -30
find $OUT -name "*.csv" -size +20M -delete
