#!/bin/bash
# same-box A/B of an environment switch on the module-API video frame: bash tools/video_api_ab.sh VAR "v1 v2" [reps=3]
VAR=$1; VALS=$2; REPS=${3:-3}
for rep in $(seq $REPS); do for v in $VALS; do
  echo -n "$VAR=$v "; env $VAR=$v timeout 300 python tools/video_api_phases.py 48 2>/dev/null | tail -1 | cut -c1-105
done; done
