"""Pivots/s of the first pivots of the BASELINE config points, measured like bench.py's config_points (2 000 pivots after 200).
usage: ab_mid_rates.py [reps]   (run from the tree whose library is to be measured: the r02 tree has its own copy)"""
import json, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1] if Path(__file__).resolve().parent.name == "scripts" else Path.cwd()
sys.path.insert(0, str(Path.cwd()))
import bench
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for wl, r in (("gridgen_8_14a", 1), ("goto_8_16a", 0), ("netgen_8_14a", 0), ("netgen_8_14a", 2), ("netgen_1m_16m", 2)):
    best = 0
    for _ in range(reps):
        m = bench.measure_single(wl, 2000, 200, r, profile_pass=False, full_sweeps=0)
        best = max(best, m["pivots_per_sec"])
    print(json.dumps({"workload": wl, "rule": r, "kpivots_s": round(best / 1e3, 2)}), flush=True)
