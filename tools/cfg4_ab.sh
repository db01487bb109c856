#!/bin/bash
# same-box A/B of an environment switch on the cfg4 leg: bash tools/cfg4_ab.sh VAR "v1 v2" [extra bench args]
VAR=$1; VALS=$2; shift 2
for rep in 1 2; do for v in $VALS; do
  env $VAR=$v timeout 400 python bench.py --workload cfg4 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$VAR=$v', d['value'], 'frames/s', d['ms_per_step'], 'ms/step, heads', d['cfg4']['heads_merge_records_ms_per_step'])"
done; done
