"""summarise a rocprofv3 --kernel-trace CSV of the video loop per FRAME: a frame starts at its first k_nhwc_ingest launch; prints the
median over frames of (span of the heads = first ingest .. last k_pan_argmax*, span of the whole frame, busy time of all kernels,
number of kernels) and the largest idle gaps inside a frame with the kernels around them.
usage: python tools/trace_gaps.py <kernel_trace.csv> [skip_frames=8]"""
import csv, sys, statistics as st
rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda t: t[0])
starts = [i for i, k in enumerate(ks) if "k_nhwc_ingest" in k[2] and (i == 0 or not any("k_nhwc_ingest" in q[2] for q in ks[max(0, i - 6):i]))]
frames = [ks[a:b] for a, b in zip(starts, starts[1:])][skip:]
heads, whole, busy, nk, period = [], [], [], [], []
gaps = {}
for f, nxt in zip(frames, frames[1:]):
    t0 = f[0][0]
    last_arg = max((k[1] for k in f if "k_pan_argmax" in k[2]), default=t0)
    heads.append((last_arg - t0) / 1e3)
    whole.append((max(k[1] for k in f) - t0) / 1e3)
    period.append((nxt[0][0] - t0) / 1e3)
    busy.append(sum(k[1] - k[0] for k in f) / 1e3)
    nk.append(len(f))
    end = f[0][1]
    for a, b in zip(f, f[1:]):
        end = max(end, a[1])
        g = (b[0] - end) / 1e3
        if g > 15:
            key = (a[2][:40], b[2][:40])
            gaps.setdefault(key, []).append(g)
med = lambda v: round(st.median(v), 1)
print({"frames": len(heads), "heads_span_us": med(heads), "frame_span_us": med(whole), "frame_period_us": med(period), "kernel_busy_sum_us": med(busy), "kernels": med(nk)})
for key, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:12]:
    print(f"  idle {st.median(v):7.1f} us x {len(v):3d}  after {key[0]:40s} before {key[1]}")
