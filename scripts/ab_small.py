"""Headline path (netgen_8_08a, LDS loop): pivots/s at bench.py's two step counts, best of `reps`.  usage: ab_small.py [reps]"""
import json, sys
from pathlib import Path
sys.path.insert(0, str(Path.cwd()))
import bench
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for steps, warm in ((20, 5), (400, 40), (2000, 100)):
    best = None
    for _ in range(reps):
        m = bench.measure_single("netgen_8_08a", steps, warm, 0, profile_pass=False)
        if best is None or m["pivots_per_sec"] > best["pivots_per_sec"]:
            best = m
    print(json.dumps({"steps": steps, "kpivots_s": round(best["pivots_per_sec"] / 1e3, 1), "us_per_step": round(1e3 * best["ms_per_step"], 2),
                      "us_in_kernel": round(best.get("roofline", {}).get("us_per_pivot_in_kernel", 0), 2)}), flush=True)
