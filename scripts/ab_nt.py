#!/usr/bin/env python3
"""Key-code sweep with / without the non-temporal load hint, below and beyond the Infinity Cache (MCF_NT_SWEEP=0/1)."""
import faulthandler
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from network_flow_solver_amd import engine, generators  # noqa: E402

faulthandler.dump_traceback_later(150, repeat=True)   # a stalled step shows where it stands
for name in ("netgen_1m_16m", "netgen_6m_96m"):
    inst = generators.named_instance(name)
    print(f"{name}: generated", flush=True)
    for nt in ("0", "1"):
        os.environ["MCF_NT_SWEEP"] = nt
        with engine.McfEngine(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=0, full_sweeps=1) as eng:
            eng.solve(max_pivots=64)
            ms = [eng.time_pricing(reps=40) for _ in range(3)]
            t0 = time.time()
            eng.solve(max_pivots=600)
            dt = time.time() - t0
            print(f"{name} nt={nt}: sweep back-to-back {[round(1e3 * x, 2) for x in ms]} us; {600 / dt:.0f} pivots/s", flush=True)
