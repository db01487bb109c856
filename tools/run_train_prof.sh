# GPU box: wall time of one training step, then its kernel mix (rocprofv3 kernel stats -> gpurun_out/train_kernel_stats.csv)
python tools/train_time.py cfg2 2 2>&1 | tail -1
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/train_prof -o t -- python $R/tools/train_time.py cfg2 2 > /tmp/train_prof.log 2>&1
mkdir -p $R/gpurun_out
F=$(find /tmp/train_prof -name '*kernel_stats.csv' | head -1)
cp "$F" $R/gpurun_out/train_kernel_stats.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/train_kernel_stats.csv")))
for r in rows[:26]: print(r["Name"][:110], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
PY
