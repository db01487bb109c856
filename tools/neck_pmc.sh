#!/bin/bash
# HBM traffic per kernel of one neck forward (16 frames, fp16 grade): separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/neck_only.py
export TMPDIR=/tmp
R=$PWD
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/np_$c
  (cd /tmp && PH_NECK_STREAMS=0 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/np_$c -o p -- python $R/tools/neck_only.py fp16 16 > /tmp/np_$c.log 2>&1 || tail -5 /tmp/np_$c.log)
done
python - <<'PY'
import csv, glob
from collections import defaultdict
def per(c):
    f = glob.glob(f"/tmp/np_{c}/**/*counter_collection.csv", recursive=True)[0]
    acc, n = defaultdict(float), defaultdict(set)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c: continue
        k = (r["Kernel_Name"][:52], r["Grid_Size"])
        acc[k] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    return {k: (acc[k] / len(n[k]), len(n[k])) for k in acc}
fe, wr = per("FETCH_SIZE"), per("WRITE_SIZE")
tot = 0
for k in sorted(fe, key=lambda k: -(2 * fe[k][0] + wr.get(k, (0, 0))[0])):
    rd, w = 2 * fe[k][0] * 1024 / 1e6, wr.get(k, (0, 0))[0] * 1024 / 1e6
    if rd + w < 20: continue
    print(f"{k[0]:52s} grid {k[1]:>8s}  launches {fe[k][1]:4d}  read {rd:7.0f} MB  written {w:7.0f} MB")
PY
