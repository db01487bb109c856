"""GPU busy / idle account of a rocprofv3 --kernel-trace CSV over its steady-state half: union of all kernel intervals, idle time, the largest
idle gaps with the kernels around them, busy time per queue.  usage: python tools/trace_busy.py <kernel_trace.csv> [top = 12]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 12
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-48:], r["Queue_Id"]) for r in rows)
t_lo = ks[len(ks) // 2][0]
ks = [k for k in ks if k[0] >= t_lo]
t0, t1 = ks[0][0], max(k[1] for k in ks)
busy, end, gaps = 0, t0, []
prev = None
for a, b, n, q in ks:
    if a > end:
        gaps.append((a - end, prev, n))
        end = a
    if b > end:
        busy += b - max(a, end) if a >= end else b - end
        end = b
        prev = n
print(f"window {(t1 - t0) / 1e6:.2f} ms, kernels {len(ks)}, GPU busy (union) {busy / (t1 - t0):.3f}, idle {(t1 - t0 - busy) / 1e6:.2f} ms")
perq = collections.defaultdict(int)
for a, b, n, q in ks:
    perq[q] += b - a
print("busy per queue (sum of durations / window):", {q: round(v / (t1 - t0), 3) for q, v in sorted(perq.items())})
agg = collections.defaultdict(lambda: [0, 0])
for g, p, n in gaps:
    agg[(p, n)][0] += g; agg[(p, n)][1] += 1
for (p, n), (g, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]:
    print(f"  idle {g / 1e3:9.1f} us total in {c:4d} gaps  after {p}  before {n}")
