#!/usr/bin/env python3
"""k_update time per pivot with and without the position-space sizes / coarse index (cycle_scan = -1 keeps neither)."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from network_flow_solver_amd import engine, generators  # noqa: E402

for name in ("netgen_1m_16m", "netgen_8_14a"):
    inst = generators.named_instance(name)
    for cs in (-1, 0):
        with engine.McfEngine(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=0, full_sweeps=1,
                              cycle_scan=cs, profile=True) as eng:
            eng.solve(max_pivots=200)
            s0 = eng.stats()
            eng.solve(max_pivots=1000)
            s1 = eng.stats()
            n = max(s1["pivot_launches"] - s0["pivot_launches"], 1)
            print(name, "cycle_scan", cs, "price/pivot/update us",
                  tuple(round(1e3 * (s1[k] - s0[k]) / n, 2) for k in ("price_ms", "pivot_ms", "apply_ms")),
                  "positions moved/pivot", round((s1["nodes_moved"] - s0["nodes_moved"]) / max(s1["pivots"] - s0["pivots"], 1)), flush=True)
