# GPU box: wall time of one training step, then its kernel mix (rocprofv3 kernel stats -> gpurun_out/train_kernel_stats.csv)
R=$PWD
OUTD0=${1:-gpurun_out}; mkdir -p $OUTD0
python tools/train_time.py cfg2 2 2>/dev/null | tail -1 | tee $OUTD0/train_step.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/train_prof -o t -- python $R/tools/train_time.py cfg2 2 > /tmp/train_prof.log 2>&1
OUTD=${1:-$R/gpurun_out}; case $OUTD in /*) ;; *) OUTD=$R/$OUTD;; esac; mkdir -p $OUTD
F=$(find /tmp/train_prof -name '*kernel_stats.csv' | head -1)
cp "$F" $OUTD/train_kernel_stats.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$OUTD/train_kernel_stats.csv")))
for r in rows[:26]: print(r["Name"][:110], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
PY
