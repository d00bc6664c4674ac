#!/usr/bin/env python3
"""Candidate list at 1 M / 16 M: slots launched per pivot made (a captured graph has a fixed sweep cadence: one sweep per
minor_cap + 1 slots; slots after the list ran dry are no-ops), with incremental and full sweeps.  python scripts/ab_inc_list.py"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from network_flow_solver_amd import engine, generators  # noqa: E402

inst = generators.named_instance("netgen_1m_16m")
for fs in (0, 1):
    with engine.McfEngine(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=2, full_sweeps=fs) as eng:
        eng.solve(max_pivots=200)
        s0 = eng.stats()
        t0 = time.time(); eng.solve(max_pivots=2000); dt = time.time() - t0
        s1 = eng.stats()
        d = {k: s1[k] - s0[k] for k in ("pivots", "pivot_launches", "price_launches", "apply_launches", "batches")}
        print(f"candidate list 1M/16M full_sweeps={fs}: {d['pivots'] / dt / 1e3:.1f} K pivots/s; {d}", flush=True)
