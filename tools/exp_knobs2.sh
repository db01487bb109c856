#!/bin/bash
Q="--no-cpu-baseline --no-kernel-head --no-neck --steps 30 --warmup 5"
for cfg in "4 96" "3 96" "6 96" "4 128" "6 144" "8 192" "4 192" "2 96" "4 64"; do
  set -- $cfg
  python bench.py $Q --streams $1 --frames $2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('streams $1 frames $2:', d['value'], d['ms_per_step'])"
done
