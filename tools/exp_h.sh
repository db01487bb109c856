Q="--no-cpu-baseline --no-kernel-head --no-neck --steps 30 --warmup 5"
for rep in 1 2; do
python bench.py $Q 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nt stores', d['value'], d['kernels_ms']['dynconv_up2_mask'], d['kernels_ms']['dynconv_up2_depth'])"
PH_ALT_LIB=tools/libpolyhead_upmcached.so python bench.py $Q 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cached stores', d['value'], d['kernels_ms']['dynconv_up2_mask'], d['kernels_ms']['dynconv_up2_depth'])"
done
