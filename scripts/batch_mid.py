#!/usr/bin/env python3
"""Throughput of BATCHES of mid-size instances (mcf_solve_batch over persistent-loop handles: one workgroup per instance,
state in global memory) against solving the same instances one after the other on the kernel-per-phase graph.
    python scripts/batch_mid.py"""
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from network_flow_solver_amd import engine, generators  # noqa: E402

for n, m in ((1024, 8192), (4096, 32768), (8192, 65536)):
    for rule in (0, 1):
        for R in ((256, 512) if os.environ.get("BATCH_MID_QUICK") else (1, 64, 256, 512)):
            if n >= 8192 and R > 256:
                continue
            insts = [generators.netgen_style(n, m, seed=1 + k) for k in range(R)]
            engines = [engine.McfEngine(i.n, i.tail, i.head, i.cost, i.cap, i.supply, rule=rule, fused=False, mid_loop=1) for i in insts]
            try:
                engine.solve_batch(engines[:1], max_pivots=5)
                engines[0].reset()
                t0 = time.time()
                ms = engine.solve_batch(engines)
                wall = time.time() - t0
                pivots = sum(eng.stats()["pivots"] for eng in engines)
                ok = all(eng.stats()["status"] == "optimal" for eng in engines)
            finally:
                for eng in engines:
                    eng.close()
            row = {"nodes": n, "arcs": m, "rule": rule, "instances": R, "pivots": pivots, "kernel_ms": round(ms, 2), "wall_ms": round(1e3 * wall, 2),
                   "pivots_per_sec_wall": round(pivots / wall), "solves_per_sec": round(R / wall, 1), "all_optimal": ok}
            if R == 64 and not os.environ.get("BATCH_MID_QUICK"):   # the same 64 instances one after the other, engine default path
                t0 = time.time()
                piv1 = 0
                for i in insts:
                    with engine.McfEngine(i.n, i.tail, i.head, i.cost, i.cap, i.supply, rule=rule, mid_loop=-1) as eng:
                        eng.solve()
                        piv1 += eng.stats()["pivots"]
                dt = time.time() - t0
                row["one_by_one_graph_path"] = {"pivots_per_sec_incl_create": round(piv1 / dt), "solves_per_sec": round(64 / dt, 1)}
            print(json.dumps(row), flush=True)
