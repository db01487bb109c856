"""Per-kernel SQ counter ratios of the rocprofv3 --pmc SQ_* passes (tools/collect_profiles.sh).
MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles summed over the SIMDs; SQ_BUSY_CYCLES is summed over the 32
shader engines (8 XCDs x 4), so the kernel's duration in cycles is SQ_BUSY_CYCLES / 32 and
    mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (duration x 1024 SIMDs) = SQ_VALU_MFMA_BUSY_CYCLES / (32 x SQ_BUSY_CYCLES);
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wave: their ratios are per resident wave."""
import csv, glob, sys
from collections import defaultdict
for src in sys.argv[1:]:
    for f in glob.glob(f"{src}/**/*counter_collection.csv", recursive=True):
        acc, n = defaultdict(lambda: defaultdict(float)), defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if not k.startswith("void k_") and not k.startswith("k_"):
                continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
        print(f"# {f}")
        for k, c in acc.items():
            d = len(n[k])
            busy, wave = c.get("SQ_BUSY_CYCLES", 0) / d, c.get("SQ_WAVE_CYCLES", 0) / d
            if not busy or not wave:
                continue
            print(f"{k[:64]:64s} launches {d:4d}  cycles/launch {busy / 32:10.0f}  mfma_busy {c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / d / (32 * busy):.3f}"
                  f"  waves waiting {c.get('SQ_WAIT_ANY', 0) / d / wave:.3f}  issue wait {c.get('SQ_WAIT_INST_ANY', 0) / d / wave:.3f}"
                  f"  active {c.get('SQ_ACTIVE_INST_ANY', 0) / d / wave:.3f}")
