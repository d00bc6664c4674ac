"""Warm-start fuzzing on the GPU box: solve, perturb supplies / costs / capacities, re-solve from the old basis
(mcf_set_basis) and compare the optimum with a cold solve of the CPU emulation.  usage: fuzz_warm.py [seconds] [first_seed]"""
import json, random, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np
import oracle
from network_flow_solver_amd import engine, generators

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 100
t_end = time.time() + budget
runs = fails = applied = fewer = 0
while time.time() < t_end:
    rng = random.Random(seed)
    fam = rng.choice(["netgen", "gridgen", "goto"])
    if fam == "netgen":
        n = rng.choice([40, 130, 300, 700, 1500, 3000]); inst = generators.netgen_style(n, n * rng.choice([4, 8]), seed=seed)
    elif fam == "gridgen":
        w = rng.choice([6, 12, 20, 33, 50]); inst = generators.gridgen_style(w, w, seed=seed)
    else:
        w = rng.choice([6, 12, 20, 33, 50]); inst = generators.goto_style(w, w, seed=seed)
    rule = rng.choice([0, 1, 2])
    opts = dict(fused=rng.random() < 0.5, mid_loop=rng.choice([-1, 0, 1]))
    with engine.McfEngine(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=rule, **opts) as eng:
        eng.solve()
        cold = eng.result()
    in_tree = cold.in_tree.astype(np.int8)
    at_upper = (~cold.in_tree & (cold.flow == inst.cap) & (inst.cap > 0)).astype(np.int8)
    supply, cost, cap = inst.supply.copy(), inst.cost.copy(), inst.cap.copy()
    kind = rng.choice(["supply", "cost", "cap", "all", "none"])
    nprng = np.random.default_rng(seed)
    if kind in ("supply", "all"):
        a, b = nprng.choice(inst.n, 2, replace=False); k = int(nprng.integers(1, 6)); supply[a] += k; supply[b] -= k
    if kind in ("cost", "all"):
        idx = nprng.choice(inst.m, max(1, inst.m // 50), replace=False); cost[idx] = np.maximum(1, cost[idx] + nprng.integers(-50, 50, idx.size))
    if kind in ("cap", "all"):
        idx = nprng.choice(inst.m, max(1, inst.m // 50), replace=False); cap[idx] = np.maximum(0, cap[idx] + nprng.integers(-20, 20, idx.size))
    ref = oracle.emul_solve(inst.n, inst.tail, inst.head, cost, cap, supply, rule=0)
    with engine.McfEngine(inst.n, inst.tail, inst.head, cost, cap, supply, rule=rule, **opts) as eng:
        ok_basis = eng.set_basis(in_tree, at_upper if rng.random() < 0.8 else None)
        eng.solve()
        r = eng.result()
    good = r.status == ref["status"] and (r.status != "optimal" or r.objective == ref["objective"])
    runs += 1
    applied += bool(ok_basis)
    fewer += bool(ok_basis and r.stats["pivots"] < cold.stats["pivots"])
    if not good:
        fails += 1
        print("MISMATCH", json.dumps({"seed": seed, "family": fam, "n": inst.n, "rule": rule, "opts": opts, "kind": kind, "applied": bool(ok_basis),
                                      "status": [r.status, ref["status"]], "objective": [int(r.objective), int(ref["objective"])]}), flush=True)
    seed += 1
    if runs % 50 == 0:
        print(f"  ... {runs} runs, {fails} mismatches", flush=True)
print(json.dumps({"runs": runs, "fails": fails, "basis_applied": applied, "fewer_pivots_than_cold": fewer}))
sys.exit(1 if fails else 0)
