#!/usr/bin/env python3
"""Throughput of BATCHES of small instances (mcf_solve_batch: one launch, one LDS-resident workgroup = one CU per
instance): R independent netgen_8_08a-sized instances (different seeds), whole solves, pivots/s over the batch.
    python scripts/batch_small.py [rule]"""
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from network_flow_solver_amd import engine, generators  # noqa: E402

rule = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rows = []
for R in (1, 16, 64, 256, 512, 1024, 2048):
    insts = [generators.netgen_style(256, 2048, seed=1 + k) for k in range(R)]
    engines = [engine.McfEngine(i.n, i.tail, i.head, i.cost, i.cap, i.supply, rule=rule) for i in insts]
    try:
        engine.solve_batch(engines[: min(R, 4)], max_pivots=5)      # warm the kernel up
        for eng in engines[: min(R, 4)]:
            eng.reset()
        t0 = time.time()
        ms = engine.solve_batch(engines)
        wall = time.time() - t0
        pivots = sum(eng.stats()["pivots"] for eng in engines)
        ok = all(eng.stats()["status"] == "optimal" for eng in engines)
        rows.append({"instances": R, "pivots": pivots, "kernel_ms": round(ms, 3), "wall_ms": round(1e3 * wall, 3),
                     "pivots_per_sec_kernel": round(pivots / (ms / 1e3)), "pivots_per_sec_wall": round(pivots / wall),
                     "solves_per_sec_wall": round(R / wall), "all_optimal": ok})
        print(json.dumps(rows[-1]), flush=True)
    finally:
        for eng in engines:
            eng.close()
