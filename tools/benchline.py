"""prints a compact summary of a bench.py JSON line read from stdin (tuning helper)"""
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else ""
line = [l for l in sys.stdin.read().splitlines() if l.startswith("{")]
if not line:
    print(tag, "NO JSON")
    sys.exit(0)
d = json.loads(line[-1])
print(tag, "fps", d["value"], "ms/step", d["ms_per_step"], {k: round(v, 3) for k, v in d["kernels_ms"].items()})
