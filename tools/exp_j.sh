A="--workload cfg4 --steps 40 --warmup 4 --no-cpu-baseline"
r() { python -c "import json; d=json.loads(open('/tmp/o.json').read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for clip in 4 8 16; do for cap in 2 3 4 8; do for early in 0 1; do
  if [ $cap -gt $clip ]; then continue; fi
  if [ $early = 1 ]; then export PH_CFG4_EARLY_BEGIN=1; else unset PH_CFG4_EARLY_BEGIN; fi
  PH_VIDEO_CLIP_BATCH=$cap python bench.py $A --clip-frames $clip > /tmp/o.json 2>/dev/null; r clip${clip}_cap${cap}_early${early}
done; done; done
