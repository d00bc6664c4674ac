#!/usr/bin/env python3
"""Differential fuzzing of the batched launches: random sets of instances (LDS-loop and persistent-loop sizes, all three
rules, key variants, budgets and resumes, both workgroup widths) through mcf_solve_batch against the CPU emulation of each
instance.   python scripts/fuzz_batch.py [seconds] [first seed]"""
import json
import os
import random
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import oracle  # noqa: E402
from network_flow_solver_amd import engine, generators  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t_end = time.time() + budget
batches = instances = fails = 0
while time.time() < t_end:
    rng = random.Random(seed)
    os.environ["MCF_BATCH_THREADS"] = rng.choice(["512", "1024"])
    count = rng.choice([1, 3, 8, 20, 40])
    jobs = []
    for k in range(count):
        n = rng.choice([30, 60, 128, 256, 300, 500, 900, 1600, 2500])
        fam = rng.choice(["netgen", "netgen", "gridgen"])
        inst = generators.netgen_style(n, n * rng.choice([4, 8]), seed=seed * 100 + k) if fam == "netgen" else generators.gridgen_style(max(4, int(n ** 0.5)), max(4, int(n ** 0.5)), seed=seed * 100 + k)
        rule = rng.choice([0, 1, 2])
        key_mode = rng.choice([0, 0, 1, 2, 3]) if rule != 1 else 0
        prio = np.random.default_rng(seed * 100 + k).integers(0, 4, size=len(inst.tail)).astype(np.int8) if key_mode == 2 else None
        jobs.append((inst, rule, key_mode, prio))
    engines = [engine.McfEngine(i.n, i.tail, i.head, i.cost, i.cap, i.supply, rule=r, mid_loop=1, key_mode=km, arc_priority=pr) for i, r, km, pr in jobs]
    try:
        cap1 = rng.choice([None, 1, 37, 400])
        if cap1 is not None:
            engine.solve_batch(engines, max_pivots=cap1)
        engine.solve_batch(engines)
        for (inst, rule, km, pr), eng in zip(jobs, engines):
            em = oracle.emul_solve(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=rule | (km << 8), arc_priority=pr)
            r, t = eng.result(), eng.tree()
            ok = (r.status == em["status"] and r.stats["pivots"] == em["pivots"] and np.array_equal(r.flow, em["flow"])
                  and np.array_equal(r.potential, em["potential"]) and np.array_equal(t["order"], em["order"]))
            instances += 1
            if not ok:
                fails += 1
                print("MISMATCH", json.dumps({"seed": seed, "n": inst.n, "rule": rule, "key_mode": km, "cap1": cap1, "width": os.environ["MCF_BATCH_THREADS"],
                                              "mode": r.stats["pricing_mode"], "pivots": [r.stats["pivots"], em["pivots"]], "status": [r.status, em["status"]]}), flush=True)
    finally:
        for eng in engines:
            eng.close()
    batches += 1
    seed += 1
    if batches % 10 == 0:
        print(f"  ... {batches} batches, {instances} instances, {fails} mismatches", flush=True)
print(json.dumps({"batches": batches, "instances": instances, "fails": fails, "last_seed": seed - 1}))
