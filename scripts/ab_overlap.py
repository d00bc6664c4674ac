#!/usr/bin/env python3
"""Overlapped graphs (pricing of pivot t+1 on a second stream beside the tree permutation of pivot t) against the
one-stream graph: pivots/s over a window of the solve, Dantzig (incremental / key codes at scale) and Devex.
    python scripts/ab_overlap.py"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from network_flow_solver_amd import engine, generators  # noqa: E402

CASES = (("netgen_8_14a", 20_000, {}), ("netgen_8_16a", 40_000, {}), ("netgen_8_18a", 40_000, {}), ("netgen_1m_16m", 40_000, {}),
         ("netgen_1m_16m", 10_000, {"full_sweeps": 1}))
for name, window, extra in CASES:
    inst = generators.named_instance(name)
    for rule in (0, 1):
        if rule == 1 and extra:
            continue
        row = []
        for ov in (-1, 1):
            with engine.McfEngine(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=rule, overlap_update=ov, **extra) as eng:
                eng.solve(max_pivots=2000)
                p0 = eng.stats()["pivots"]
                t0 = time.time()
                eng.solve(max_pivots=window)
                dt = time.time() - t0
                st = eng.stats()
                row.append(f"overlap={ov}: {(st['pivots'] - p0) / dt / 1e3:.1f} K pivots/s")
        print(f"{name} rule={rule} {extra}: " + " | ".join(row), flush=True)
