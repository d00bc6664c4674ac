#!/usr/bin/env python3
"""Crossover of the three engine modes on small and mid-size instances: whole solves with the LDS loop (mode 2), the
persistent single-workgroup loop (mode 3) and the kernel-per-phase graph (mode 1) forced in turn.  python scripts/ab_modes.py"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from network_flow_solver_amd import engine, generators  # noqa: E402

SIZES = ((256, 2048), (512, 4096), (1024, 8192), (2048, 16384), (4096, 32768), (8192, 65536))
FORCE = (("lds loop", dict(fused=True, mid_loop=-1)), ("mid loop", dict(fused=False, mid_loop=1)), ("graph", dict(fused=False, mid_loop=-1)))
for n, m in SIZES:
    inst = generators.netgen_style(n, m, seed=1)
    for rule in (0, 1, 2):
        row = []
        for label, kw in FORCE:
            best = None
            for _ in range(2):
                with engine.McfEngine(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=rule, **kw) as eng:
                    t0 = time.time()
                    eng.solve()
                    dt = time.time() - t0
                    st = eng.stats()
                if best is None or dt < best[0]:
                    best = (dt, st)
            dt, st = best
            row.append(f"{label}: mode {st['pricing_mode']} {st['pivots']} pivots {1e3 * dt:.1f} ms = {st['pivots'] / dt / 1e3:.1f} K/s")
        print(f"netgen {n} / {m} rule={rule}: " + " | ".join(row), flush=True)
