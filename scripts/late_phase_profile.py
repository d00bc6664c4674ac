#!/usr/bin/env python3
"""Per-kernel time per pivot EARLY and LATE in a large solve: run `skip` pivots at full speed, move the basis into a
profiled handle (HIP events around every kernel, eager launches) by warm start, time the next `span` pivots.
    python scripts/late_phase_profile.py [instance] [rule] [skip...]"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from network_flow_solver_amd import engine, generators  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "netgen_1m_16m"
rule = int(sys.argv[2]) if len(sys.argv) > 2 else 2
skips = [int(x) for x in sys.argv[3:]] or [0, 1_000_000, 2_000_000, 3_000_000]
span = 4000
inst = generators.named_instance(name)
fast = engine.McfEngine(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=rule)
done = 0
for skip in skips:
    t0 = time.time()
    if skip > done:
        fast.solve(max_pivots=skip - done)
        done = skip
    res = fast.result()
    if res.status == "optimal":
        print(f"{name}: optimal after {res.stats['pivots']} pivots", flush=True)
        break
    at_upper = ~res.in_tree & (inst.cap > 0) & (res.flow == inst.cap)
    with engine.McfEngine(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=rule, profile=True) as prof:
        ok = prof.set_basis(res.in_tree, at_upper) if skip else True
        s0 = prof.stats()
        prof.solve(max_pivots=span)
        s1 = prof.stats()
        n = max(s1["pivot_launches"] - s0["pivot_launches"], 1)
        piv = s1["pivots"] - s0["pivots"]
        print(f"{name} rule={rule} after {skip} pivots (basis moved: {ok}): per pivot slot  price {1e3 * (s1['price_ms'] - s0['price_ms']) / n:.2f} us  "
              f"pivot {1e3 * (s1['pivot_ms'] - s0['pivot_ms']) / n:.2f} us  update {1e3 * (s1['apply_ms'] - s0['apply_ms']) / n:.2f} us  "
              f"| cycle arcs/pivot {(s1['cycle_arcs'] - s0['cycle_arcs']) / max(piv, 1):.1f}  positions moved/pivot "
              f"{(s1['nodes_moved'] - s0['nodes_moved']) / max(piv, 1):.0f}  subtree nodes/pivot {(s1['subtree_nodes'] - s0['subtree_nodes']) / max(piv, 1):.0f}  "
              f"scan rounds/scan {(s1['scan_rounds'] - s0['scan_rounds']) / max(s1['cycle_scans'] - s0['cycle_scans'], 1):.2f}  ({time.time() - t0:.0f} s)", flush=True)
fast.close()
