#!/bin/bash
# SQ / LDS counters of k_conv_nhwc (tools/conv_ab.py: 3x3 stride 1 and 2 at cfg2's sizes, 16 frames, fp16 grade).  usage: bash tools/conv_pmc.sh [lib]
export TMPDIR=/tmp
R=$PWD; L=${1:--}
rm -rf /tmp/pc1 /tmp/pc2; mkdir -p gpurun_out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pc1 -o p -- python tools/conv_ab.py $L > /dev/null 2>&1
python tools/pmc_sq_summary.py /tmp/pc1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d /tmp/pc2 -o p -- python tools/conv_ab.py $L > /dev/null 2>&1
python - <<PY
import csv, glob
from collections import defaultdict
for f in glob.glob("/tmp/pc2/**/*counter_collection.csv", recursive=True):
    acc, n = defaultdict(lambda: defaultdict(float)), defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_conv" not in k: continue
        key = k[:34] + " grid " + r.get("Grid_Size", "?")
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); n[key].add(r["Dispatch_Id"])
    for k, c in acc.items():
        d = len(n[k]); w = c["SQ_WAVE_CYCLES"] / d
        print(k, "launches", d, " ".join(f"{kk[3:]}={v / d / w:.3f}" for kk, v in c.items() if kk != "SQ_WAVE_CYCLES"), "(per wave cycle)")
PY
