#!/usr/bin/env python3
"""Cycle search at 1 M nodes: climb k round trips then scan (coarse index), pivots/s early and late in the solve.
    python scripts/ab_cycle_big.py [instance] [rule]"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from network_flow_solver_amd import engine, generators  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "netgen_1m_16m"
rule = int(sys.argv[2]) if len(sys.argv) > 2 else 2
inst = generators.named_instance(name)
for cs in (1, 3, 5, 9, 17):
    with engine.McfEngine(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=rule, cycle_scan=cs) as eng:
        rates = []
        for leg in range(4):
            t0 = time.time()
            eng.solve(max_pivots=500_000)
            rates.append(500_000 / (time.time() - t0))
        st = eng.stats()
        print(f"{name} rule={rule} cycle_scan={cs}: pivots/s per 500K-pivot leg {[round(r) for r in rates]}  scans={st['cycle_scans']} "
              f"cycle_arcs/pivot={st['cycle_arcs'] / st['pivots']:.1f}", flush=True)
