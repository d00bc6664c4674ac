#!/usr/bin/env python3
"""Grid size of the key-code sweep: back-to-back launch time for 256 ... 2048 pricing workgroups (price_blocks).
    python scripts/ab_price_blocks.py"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from network_flow_solver_amd import engine, generators  # noqa: E402

for name in ("netgen_1m_16m", "netgen_6m_96m"):
    inst = generators.named_instance(name)
    for pb in (256, 512, 1024, 2048):
        with engine.McfEngine(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=0, full_sweeps=1, price_blocks=pb) as eng:
            eng.solve(max_pivots=64)
            ms = sorted(eng.time_pricing(reps=40) for _ in range(3))
            print(f"{name} price_blocks={pb}: sweep back-to-back {1e3 * ms[0]:.2f} us (best of 3; {1e3 * ms[-1]:.2f} worst)", flush=True)
