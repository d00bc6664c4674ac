#!/usr/bin/env python3
"""A/B of the compressed Dantzig keys: per-kernel time per pivot (HIP events, profiled pass) and pivots/s of the normal
path, with and without the 4-byte key codes.  python scripts/ab_keys.py"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from network_flow_solver_amd import engine, generators  # noqa: E402

for name, rule, fs, steps in (("netgen_8_14a", 0, 1, 4000), ("goto_8_16a", 0, 1, 4000), ("netgen_1m_16m", 0, 1, 1000),
                              ("netgen_1m_16m", 0, 0, 4000), ("netgen_1m_16m", 2, 0, 4000)):
    inst = generators.named_instance(name)
    for ck in (-1, 1):
        row = {}
        for profile in (False, True):
            with engine.McfEngine(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=rule, full_sweeps=fs,
                                  compressed_keys=ck, profile=profile) as eng:
                eng.solve(max_pivots=200)
                s0 = eng.stats()
                t0 = time.time()
                eng.solve(max_pivots=steps if not profile else min(steps, 1000))
                dt = time.time() - t0
                s1 = eng.stats()
                if profile:
                    n = max(s1["pivot_launches"] - s0["pivot_launches"], 1)
                    row["us"] = tuple(round(1e3 * (s1[k] - s0[k]) / n, 2) for k in ("price_ms", "pivot_ms", "apply_ms"))
                else:
                    row["pivots_per_s"] = round((s1["pivots"] - s0["pivots"]) / dt)
        print(f"{name} rule={rule} full_sweeps={fs} compressed_keys={ck}: {row['pivots_per_s']} pivots/s; price/pivot/update us {row['us']}", flush=True)
