export TMPDIR=/tmp
OUT=gpurun_out/r06i; mkdir -p $OUT
A="--workload cfg4 --steps 30 --warmup 4 --clip-frames 8 --no-cpu-baseline"
rocprofv3 --kernel-trace --output-format csv -d $OUT/late -o t -- python bench.py $A > $OUT/late.json 2> $OUT/late.err
PH_CFG4_EARLY_BEGIN=1 rocprofv3 --kernel-trace --output-format csv -d $OUT/early -o t -- python bench.py $A > $OUT/early.json 2> $OUT/early.err
for v in late early; do echo "== $v"; python -c "import json; d=json.loads(open('$OUT/$v.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; python tools/trace_busy.py $(find $OUT/$v -name "*kernel_trace.csv") 14; done
find $OUT -name "*.csv" -size +20M -delete
