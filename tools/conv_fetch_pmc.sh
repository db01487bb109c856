#!/bin/bash
# FETCH_SIZE of the neck's conv launches (tools/conv_ab.py: 3x3 s1 @128x256, 3x3 s2 @256x512, 3x3 s1 @64x128, 16 frames) with channels-last
# and with chunk-major planes into the stride-2 kernel: two rocprofv3 --pmc passes (counters only + kernel trace).  bash tools/conv_fetch_pmc.sh
export TMPDIR=/tmp
R=$PWD
for lay in nhwc c16; do
  rm -rf /tmp/cf_$lay
  (cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/cf_$lay -o p -- python $R/tools/conv_ab.py - 16 $lay > /tmp/cf_$lay.log 2>&1 || tail -5 /tmp/cf_$lay.log)
  python - <<PY
import csv, glob
from collections import defaultdict
f = glob.glob("/tmp/cf_$lay/**/*counter_collection.csv", recursive=True)[0]
acc, n = defaultdict(float), defaultdict(set)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != "FETCH_SIZE" or "k_conv_nhwc" not in r["Kernel_Name"]:
        continue
    key = (r["Kernel_Name"][:44], r["Grid_Size"])
    acc[key] += float(r["Counter_Value"]); n[key].add(r["Dispatch_Id"])
for k in acc:
    kib = acc[k] / len(n[k])
    print("$lay", k[0], "grid", k[1], f"FETCH_SIZE {kib / 1024:.0f} MiB per launch -> {2 * kib * 1024 / 1e6:.0f} MB of reads (gfx950: 2 x FETCH_SIZE)")
PY
done
echo "inputs per launch (16 frames, fp16): 128x256 plane 268 MB, 256x512 plane 1074 MB, 64x128 plane 67 MB; weights 1.2 MB"
