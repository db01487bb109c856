#!/bin/bash
# kernel-time shares of the cfg4 leg at 8-frame clips from a rocprofv3 --kernel-trace (DESIGN 7b "Round 6"): busy / idle account + per-kernel totals
export TMPDIR=/tmp; OUT=gpurun_out/c4t; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o t -- python bench.py --workload cfg4 --steps 30 --warmup 4 --clip-frames 8 --no-cpu-baseline > $OUT/bench.json 2> $OUT/err.log
python tools/trace_busy.py $(find $OUT/t -name "*kernel_trace.csv") 14
python - <<'P'
import csv, glob, collections
f = glob.glob("gpurun_out/c4t/t/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows)//2:]
t0, t1 = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
agg = collections.defaultdict(lambda: [0, 0])
for r in rows:
    n = r["Kernel_Name"].split("(")[0][-60:]
    agg[n][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); agg[n][1] += 1
tot = sum(v[0] for v in agg.values())
print("window ms", (t1 - t0) / 1e6, "sum of kernel time ms", tot / 1e6)
for n, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:40]:
    print(f"{t / tot * 100:5.1f} %  {t / 1e3 / c:8.1f} us x {c:5d}  {n}")
P
