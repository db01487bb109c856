#!/bin/bash
# HBM traffic per kernel of the training step (tools/train_time.py cfg2 2): separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes;
# prints MB per launch and per step beside each kernel's share of the step's GPU time is in profiles/*/train_kernel_stats.csv
export TMPDIR=/tmp
R=$PWD
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/tp_$c
  (cd /tmp && rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/tp_$c -o p -- python $R/tools/train_time.py cfg2 2 > /tmp/tp_$c.log 2>&1 || tail -5 /tmp/tp_$c.log)
done
python - <<'PY'
import csv, glob
from collections import defaultdict
def per(c):
    f = glob.glob(f"/tmp/tp_{c}/**/*counter_collection.csv", recursive=True)[0]
    acc, n = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c: continue
        k = r["Kernel_Name"][:60]
        acc[k] += float(r["Counter_Value"]); n[k] += 1
    return acc, n
(fe, nf), (wr, nw) = per("FETCH_SIZE"), per("WRITE_SIZE")
tot_r = sum(fe.values()) * 2 * 1024 / 1e6; tot_w = sum(wr.values()) * 1024 / 1e6
print(f"all launches of the run: read {tot_r:.0f} MB, written {tot_w:.0f} MB")
for k in sorted(fe, key=lambda k: -(2 * fe[k] + wr.get(k, 0)))[:22]:
    print(f"{k:60s} launches {nf[k]:5d}  read {2 * fe[k] * 1024 / 1e6:8.0f} MB  written {wr.get(k, 0) * 1024 / 1e6:8.0f} MB  ({2 * fe[k] * 1024 / 1e6 / nf[k]:6.1f} / {wr.get(k, 0) * 1024 / 1e6 / max(nw.get(k, 1), 1):6.1f} per launch)")
PY
