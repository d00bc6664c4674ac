#!/usr/bin/env python3
"""Where does the persistent single-workgroup loop (k_solve_mid, mode 3) stop paying?  Whole solves with the loop forced
on (mid_loop=1) and off (mid_loop=-1) on instances around and above the 8 192-node limit.  python scripts/ab_mid.py"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from network_flow_solver_amd import engine, generators  # noqa: E402

CASES = (("netgen_8_12a", (1, 2)), ("netgen_8_14a", (1, 2)), ("gridgen_8_14a", (1, 2)), ("goto_8_14a", (1, 2)), ("netgen_8_16a", (1, 2)))
for name, rules in CASES:
    try:
        inst = generators.named_instance(name)
    except Exception as exc:  # noqa: BLE001
        print(f"{name}: {exc}", flush=True)
        continue
    for rule in rules:
        row = []
        for mid in (-1, 1):
            with engine.McfEngine(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=rule, mid_loop=mid) as eng:
                t0 = time.time()
                eng.solve()
                dt = time.time() - t0
                st = eng.stats()
                row.append(f"mid_loop={mid}: mode {st['pricing_mode']} {st['pivots']} pivots {dt:.3f} s = {st['pivots'] / dt / 1e3:.1f} K/s ({st['status']})")
        print(f"{name} ({inst.n} nodes / {len(inst.tail)} arcs) rule={rule}: " + " | ".join(row), flush=True)
