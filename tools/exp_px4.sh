#!/bin/bash
export TMPDIR=/tmp; OUT=gpurun_out/px; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_sq -o p -- python tools/px_time.py > /dev/null 2> $OUT/pmc_sq.err
python tools/pmc_sq_summary.py $OUT/pmc_sq
python - <<'P'
import csv, glob
from collections import defaultdict
for f in glob.glob("gpurun_out/px/pmc_sq/**/*counter_collection.csv", recursive=True):
    acc, n = defaultdict(lambda: defaultdict(float)), defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k, c in acc.items():
        d = len(n[k])
        print(k, "LDS active", c["SQ_ACTIVE_INST_LDS"]/d, "bank conflict", c["SQ_LDS_BANK_CONFLICT"]/d, "ratio", c["SQ_LDS_BANK_CONFLICT"]/max(1,c["SQ_ACTIVE_INST_LDS"]))
P
