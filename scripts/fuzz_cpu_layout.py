#!/usr/bin/env python3
"""CPU-only fuzz: the emulation on the blocked preorder list (random block size, random pool incl. none) against the
emulation on the dense preorder array -- pivots, flows, potentials, order, positions, sizes, depths must be identical.
    python scripts/fuzz_cpu_layout.py [seconds] [first_seed]"""
import json, random, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, oracle
from network_flow_solver_amd import generators
t_end = time.time() + (float(sys.argv[1]) if len(sys.argv) > 1 else 240)
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
runs = bad = 0
while time.time() < t_end:
    rng = random.Random(seed)
    n = rng.choice([12, 40, 90, 250, 600, 1500])
    fam = rng.choice(["netgen", "gridgen", "goto"])
    if fam == "netgen": inst = generators.netgen_style(n, n * rng.choice([3, 6, 10]), seed=seed)
    elif fam == "gridgen": w = max(3, int(n ** 0.5)); inst = generators.gridgen_style(w, w + rng.choice([0, 2]), seed=seed)
    else: w = max(3, int(n ** 0.5)); inst = generators.goto_style(w, w, seed=seed)
    rule = rng.choice([0, 1, 2])
    cb = rng.choice([-1, 0, 2])
    dense = oracle.emul_solve(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=rule, climb_budget=cb)
    shift = rng.choice([2, 3, 4, 5, 6])
    pool = rng.choice([0, 1, 2, 3, 9, 40])          # decode_rule: 0 auto, 1 none, k -> k - 1 spare blocks
    bits = rule | (shift << 16) | (pool << 20)
    blk = oracle.emul_solve(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=bits, climb_budget=cb)
    ok = (dense["status"] == blk["status"] and dense["pivots"] == blk["pivots"] and dense["objective"] == blk["objective"]
          and all(np.array_equal(dense[k], blk[k]) for k in ("flow", "potential", "order", "pos", "psize", "depth", "parent")))
    runs += 1
    if not ok:
        bad += 1
        print("MISMATCH", json.dumps({"seed": seed, "family": fam, "n": inst.n, "rule": rule, "shift": shift, "pool": pool, "climb_budget": cb,
                                      "pivots": [dense["pivots"], blk["pivots"]]}), flush=True)
    seed += 1
print(json.dumps({"runs": runs, "bad": bad, "last_seed": seed - 1}))
sys.exit(1 if bad else 0)
