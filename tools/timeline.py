"""Overlap timeline of the multi-stream decode step from a rocprofv3 --kernel-trace CSV of `python bench.py` (the graph replays its
parts on as many hardware queues): over the steady-state replays, per step --

  wall                        period between consecutive replays (first binarize of a replay to the first of the next)
  hbm_union                   time during which at least one HBM-bound kernel (binarize, pool, dynconv, dynconv_up2) is running
  query_exposed               time during which ONLY query kernels (k_query_pre2 / post2) are running
  idle                        time during which no kernel of the step is running
  concurrency                 share of the wall time with n kernels of the step in flight
  per kernel class            launches per step, mean duration inside the step (stretched by the overlap)

and, given the single-stream trace of one part (`--isolated <csv>`: bench.py --streams 1 --frames <frames per part>), the sum of the
HBM-bound launches' isolated durations x parts = the step a perfect overlap would take, and
  overlap_efficiency = that sum / wall.

usage: python tools/timeline.py <kernel_trace.csv> [--isolated <kernel_trace.csv>] [--skip 14] [--json out.json]"""
import argparse
import csv
import json
import statistics as st

HBM = ("k_binarize", "k_pool", "k_dynconv_up2", "k_dynconv", "k_upsample2x", "k_ingest")
QUERY = ("k_query_pre", "k_query_post")


def klass(name, grid_y=None):
    n = name.split("(")[0]
    if "k_dynconv_poolx" in n:                  # round 6: non-final conv + pooling of the x map
        return "dynconv_poolx"
    if "k_pool" in n and grid_y == 2:           # k_pool over ONE map (depth_feats; the x half came from k_dynconv_poolx)
        return "pool_depth"
    if "k_dynconv_up2" in n:
        return "dynconv_up2_mask" if ", true, " in n else "dynconv_up2_depth"
    if "k_dynconv" in n:
        return "dynconv_bits" if ", true, float" in n else "dynconv_logits"
    for k in HBM + QUERY:
        if k in n:
            return k[2:]
    return None


def load(path):
    ks = []
    for r in csv.DictReader(open(path)):
        c = klass(r["Kernel_Name"], int(r.get("Grid_Size_Y") or 0))
        if c:
            ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), c, r["Queue_Id"]))
    ks.sort()
    return ks


def union(iv):
    """total length of the union of intervals"""
    tot, end = 0, None
    for a, b in sorted(iv):
        if end is None or a > end:
            tot += b - a
            end = b
        elif b > end:
            tot += b - end
            end = b
    return tot


def sweep(ks, t0, t1):
    """time with n kernels in flight, time with only query kernels in flight, inside [t0, t1)"""
    ev = []
    for a, b, c, _ in ks:
        a, b = max(a, t0), min(b, t1)
        if a < b:
            q = c.startswith("query")
            ev.append((a, 1, q))
            ev.append((b, -1, q))
    ev.sort()
    conc, qonly, n, nq, last = {}, 0, 0, 0, t0
    for t, d, q in ev:
        conc[n] = conc.get(n, 0) + t - last
        if n > 0 and n == nq:
            qonly += t - last
        last = t
        n += d
        nq += d if q else 0
    conc[n] = conc.get(n, 0) + t1 - last
    return conc, qonly


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--isolated")
    ap.add_argument("--skip", type=int, default=14, help="replays to skip at the start (set-up + warm-up)")
    ap.add_argument("--json")
    ap.add_argument("--frames-per-step", type=int, default=128, help="recorded in the summary (bench.py quotes it only for that step size)")
    a = ap.parse_args()
    ks = load(a.trace)
    # the graph replays run on the queues that carry binarize launches; a replay starts with the first binarize after a previous
    # replay's last kernel class (dynconv_up2_depth / upsample2x) on the same queue
    queues = sorted({k[3] for k in ks if k[2] == "binarize"})
    if not queues:
        raise SystemExit(f"{a.trace}: no k_binarize launches -- not a trace of the decode step (did the traced command fail?)")
    parts = len(queues)
    q0 = queues[0]
    starts = [k[0] for k in ks if k[2] == "binarize" and k[3] == q0]
    # keep the dense run of replays: periods within 1.5x of the median
    per = [b - a for a, b in zip(starts, starts[1:])]
    med = st.median(per)
    dense = [i for i, p in enumerate(per) if p < 1.5 * med]
    lo, hi = dense[0] + a.skip, dense[-1]
    if hi - lo < 4:
        lo = dense[0]
    t0, t1 = starts[lo], starts[hi]
    nsteps = hi - lo
    win = [k for k in ks if k[1] > t0 and k[0] < t1]
    wall = (t1 - t0) / nsteps
    clip = lambda k: (max(k[0], t0), min(k[1], t1))
    hbm_u = union([clip(k) for k in win if not k[2].startswith("query")]) / nsteps
    all_u = union([clip(k) for k in win]) / nsteps
    conc, qonly = sweep(win, t0, t1)
    res = {"trace": a.trace, "frames_per_step": a.frames_per_step, "parts": parts, "steps_analysed": nsteps, "wall_us_per_step": round(wall / 1e3, 1),
           "hbm_union_us": round(hbm_u / 1e3, 1), "query_exposed_us": round(qonly / nsteps / 1e3, 1),
           "idle_us": round((wall - all_u) / 1e3, 1),
           "concurrency_share": {str(n): round(v / (t1 - t0), 4) for n, v in sorted(conc.items())}}
    cls = {}
    for k in win:
        if t0 <= k[0] < t1:
            cls.setdefault(k[2], []).append(k[1] - k[0])
    res["in_step"] = {c: {"launches_per_step": round(len(v) / nsteps, 2), "mean_us": round(st.mean(v) / 1e3, 1),
                          "sum_us_per_step": round(sum(v) / nsteps / 1e3, 1)} for c, v in sorted(cls.items())}
    if a.isolated:
        iso = {}
        for k in load(a.isolated):
            iso.setdefault(k[2], []).append(k[1] - k[0])
        # per class: median isolated duration x launches per step of the overlapped run
        imed = {c: st.median(v) for c, v in iso.items()}
        hbm_sum = sum(imed[c] * len(v) / nsteps for c, v in cls.items() if not c.startswith("query") and c in imed)
        q_sum = sum(imed[c] * len(v) / nsteps for c, v in cls.items() if c.startswith("query") and c in imed)
        res["isolated"] = {"trace": a.isolated, "median_us": {c: round(v / 1e3, 1) for c, v in sorted(imed.items())},
                           "hbm_bound_sum_us_per_step": round(hbm_sum / 1e3, 1), "query_sum_us_per_step": round(q_sum / 1e3, 1)}
        res["overlap_efficiency"] = round(hbm_sum / wall, 4)
        res["query_time_not_hidden_us_per_step"] = round((wall - hbm_sum) / 1e3, 1)
        res["query_time_not_hidden_us_per_part"] = round((wall - hbm_sum) / 1e3 / parts, 1)
    s = json.dumps(res, indent=1)
    print(s)
    if a.json:
        open(a.json, "w").write(s + "\n")


if __name__ == "__main__":
    main()
