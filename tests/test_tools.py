"""The measurement tools that turn rocprofv3 CSVs into the committed summaries (profiles/r06/timeline_4streams.json, pmc_traffic.json) on
tiny synthetic traces: launch classes (round 6: the fused conv + pooling kernel, k_pool over one map), the overlap account, the order rule
that tells the two k_pool forms apart in a counter CSV."""
import csv
import json
import os
import subprocess
import sys

import helpers as Hh

TOOLS = os.path.join(Hh.REPO, "tools")
K = {"bin": "void k_binarize<unsigned short, 1>(void const*, int)",
     "pool": "void k_pool<1, 5, 0>(unsigned short const*, unsigned short const*)",
     "px": "void k_dynconv_poolx<2, 5, true>(unsigned short const*, unsigned short const*)",
     "up_m": "void k_dynconv_up2m<2, 5, 4, true, ph_h16>(unsigned short const*)",
     "up_d": "void k_dynconv_up2m<2, 5, 4, false, ph_h16>(unsigned short const*)",
     "qpre": "void k_query_pre2<2, 2, true>(QArgs)", "qpost": "void k_query_post2<1, 2, 1>(QArgs)"}


def _trace(path, parts, steps, gap=0):
    """every part on its own queue, one phase (100 us) behind the previous; kernels 100 us each; a step = 10 phases"""
    rows, did = [], 0
    seq = [("bin", 4), ("pool", 4), ("qpre", 1), ("qpost", 1), ("px", 1), ("pool", 2), ("qpre", 1), ("qpost", 1), ("up_m", 1), ("up_d", 1)]
    for s in range(steps):
        for p in range(parts):
            t = (s * (len(seq) + parts - 1 + gap) + p) * 100_000
            for name, gy in seq:
                did += 1
                rows.append({"Kind": "KERNEL_DISPATCH", "Agent_Id": "Agent 2", "Queue_Id": str(p + 1), "Stream_Id": "0", "Thread_Id": "1",
                             "Dispatch_Id": str(did), "Kernel_Id": "1", "Kernel_Name": K[name], "Correlation_Id": str(did),
                             "Start_Timestamp": str(t), "End_Timestamp": str(t + 100_000), "LDS_Block_Size": "0", "Scratch_Size": "0",
                             "VGPR_Count": "0", "Accum_VGPR_Count": "0", "SGPR_Count": "0", "Workgroup_Size_X": "256", "Workgroup_Size_Y": "1",
                             "Workgroup_Size_Z": "1", "Grid_Size_X": "2560", "Grid_Size_Y": str(gy), "Grid_Size_Z": "32"})
                t += 100_000
    with open(path, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0]))
        w.writeheader()
        w.writerows(rows)


def test_timeline_classes_and_overlap_account(tmp_path):
    multi, single, out = tmp_path / "d.csv", tmp_path / "e.csv", tmp_path / "t.json"
    _trace(multi, parts=4, steps=12)
    _trace(single, parts=1, steps=12)
    subprocess.check_call([sys.executable, os.path.join(TOOLS, "timeline.py"), str(multi), "--isolated", str(single), "--skip", "2",
                           "--json", str(out), "--frames-per-step", "128"], stdout=subprocess.DEVNULL)
    t = json.load(open(out))
    assert t["parts"] == 4 and t["frames_per_step"] == 128
    assert set(t["in_step"]) == {"binarize", "pool", "pool_depth", "dynconv_poolx", "dynconv_up2_mask", "dynconv_up2_depth", "query_pre", "query_post"}
    assert t["in_step"]["pool_depth"]["launches_per_step"] == 4.0 and t["in_step"]["pool"]["launches_per_step"] == 4.0
    assert t["wall_us_per_step"] == 1300.0                      # 10 phases + the skew of three more parts
    assert t["isolated"]["median_us"]["dynconv_poolx"] == 100.0
    # six HBM-bound launches of 100 us per part x 4 parts against a 1 300 us step
    assert t["isolated"]["hbm_bound_sum_us_per_step"] == 2400.0 and abs(t["overlap_efficiency"] - 2400.0 / 1300.0) < 1e-3
    assert t["idle_us"] == 0.0


def test_pmc_summary_tells_the_two_pool_forms_apart_by_launch_order(tmp_path):
    rows, did = [], 0
    for it in range(3):
        for name, fetch, write in (("px", 200.0, 70.0), ("pool", 210.0, 38.0), ("pool", 420.0, 38.0), ("up_m", 220.0, 1100.0)):
            did += 1
            for kind, val in (("pmc_fetch", fetch), ("pmc_write", write)):
                rows.append((kind, {"Correlation_Id": did, "Dispatch_Id": did, "Agent_Id": "Agent 2", "Queue_Id": 1, "Process_Id": 1, "Thread_Id": 1,
                                    "Grid_Size": 122880, "Kernel_Id": 1, "Kernel_Name": K[name], "Workgroup_Size": 256, "LDS_Block_Size": 0,
                                    "Scratch_Size": 0, "VGPR_Count": 0, "Accum_VGPR_Count": 0, "SGPR_Count": 0,
                                    "Counter_Name": "FETCH_SIZE" if kind == "pmc_fetch" else "WRITE_SIZE", "Counter_Value": val,
                                    "Start_Timestamp": did, "End_Timestamp": did + 1}))
    for kind in ("pmc_fetch", "pmc_write"):
        d = tmp_path / kind
        d.mkdir()
        sel = [r for k, r in rows if k == kind]
        with open(d / "p_counter_collection.csv", "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(sel[0]))
            w.writeheader()
            w.writerows(sel)
    out = tmp_path / "traffic.json"
    subprocess.check_call([sys.executable, os.path.join(TOOLS, "pmc_summary.py"), str(tmp_path), str(out)], stdout=subprocess.DEVNULL,
                          env=dict(os.environ, PH_PART_FRAMES="32"))
    t = json.load(open(out))
    k = t["kernels"]
    assert t["frames_per_launch"] == 32
    assert k["pool_depth"]["FETCH_SIZE_KiB"] == 210.0 and k["pool"]["FETCH_SIZE_KiB"] == 420.0          # alternating: depth alone first
    assert k["dynconv_poolx"]["hbm_bytes_per_launch"] == int((2 * 200.0 + 70.0) * 1024)
    assert k["dynconv_up2_mask"]["WRITE_SIZE_KiB"] == 1100.0
