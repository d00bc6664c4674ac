#!/usr/bin/env python3
"""Per-kernel time per pivot over the course of a large solve: ONE profiled handle (HIP events around every kernel, eager
launches) runs the whole solve; every `window` pivots the averages of that window are printed.
    python scripts/late_phase_profile.py [instance] [rule] [window] [max_pivots]"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from network_flow_solver_amd import engine, generators  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "netgen_1m_16m"
rule = int(sys.argv[2]) if len(sys.argv) > 2 else 2
window = int(sys.argv[3]) if len(sys.argv) > 3 else 250_000
max_pivots = int(sys.argv[4]) if len(sys.argv) > 4 else 10_000_000
inst = generators.named_instance(name)
KEYS = ("price_ms", "pivot_ms", "apply_ms", "pivot_launches", "pivots", "cycle_arcs", "nodes_moved", "subtree_nodes", "scan_rounds", "cycle_scans")
with engine.McfEngine(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=rule, profile=True) as eng:
    prev = {k: eng.stats()[k] for k in KEYS}
    t_start = time.time()
    while prev["pivots"] < max_pivots:
        t0 = time.time()
        eng.solve(max_pivots=window)
        st = eng.stats()
        cur = {k: st[k] for k in KEYS}
        d = {k: cur[k] - prev[k] for k in KEYS}
        n, piv = max(d["pivot_launches"], 1), max(d["pivots"], 1)
        print(f"{name} rule={rule} pivots {prev['pivots']}..{cur['pivots']}: per pivot slot  price {1e3 * d['price_ms'] / n:.2f} us  "
              f"pivot {1e3 * d['pivot_ms'] / n:.2f} us  update {1e3 * d['apply_ms'] / n:.2f} us  | cycle arcs/pivot {d['cycle_arcs'] / piv:.1f}  "
              f"positions moved/pivot {d['nodes_moved'] / piv:.0f}  subtree nodes/pivot {d['subtree_nodes'] / piv:.0f}  "
              f"scan rounds/scan {d['scan_rounds'] / max(d['cycle_scans'], 1):.2f}  ({time.time() - t0:.0f} s, eager + events)", flush=True)
        prev = cur
        if st["status"] in ("optimal", "unbounded"):
            break
    res = eng.result()
    print(f"{name} rule={rule}: {res.status} after {res.stats['pivots']} pivots, objective {res.objective}, {time.time() - t_start:.0f} s profiled", flush=True)
