#!/usr/bin/env python3
"""Pivots per captured graph (batch_pivots): whole-solve rate on mid-size instances.  python scripts/ab_batch_pivots.py"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from network_flow_solver_amd import engine, generators  # noqa: E402

for name, rule in (("netgen_8_14a", 0), ("gridgen_8_14a", 1), ("netgen_8_14a", 2), ("netgen_8_16a", 0)):
    inst = generators.named_instance(name)
    row = []
    for bp in (32, 64, 128, 256):
        best = None
        for _ in range(2):
            with engine.McfEngine(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=rule, batch_pivots=bp) as eng:
                t0 = time.time(); eng.solve(); dt = time.time() - t0
                piv = eng.stats()["pivots"]
            best = dt if best is None or dt < best else best
        row.append(f"{bp}: {piv / best / 1e3:.1f} K/s")
    print(f"{name} rule={rule}: " + " | ".join(row), flush=True)
