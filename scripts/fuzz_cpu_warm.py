#!/usr/bin/env python3
"""CPU-only warm-start fuzz: the emulation solves, the instance is perturbed (supplies / costs / capacities), the emulation
re-solves from the old basis (mcf_apply_basis: the host code the engine shares) and must reach the optimum of the PINNED
oracle (oracle/ref_simplex.c) with a valid certificate; unperturbed, one-component bases must be confirmed in zero pivots.
    python scripts/fuzz_cpu_warm.py [seconds] [first_seed]"""
import json, random, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, oracle
from network_flow_solver_amd import generators
from conftest import check_optimality
t_end = time.time() + (float(sys.argv[1]) if len(sys.argv) > 1 else 240)
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 300
runs = bad = applied = zero = 0
while time.time() < t_end:
    rng = random.Random(seed)
    n = rng.choice([20, 60, 150, 400, 900])
    fam = rng.choice(["netgen", "gridgen", "goto"])
    if fam == "netgen": inst = generators.netgen_style(n, n * rng.choice([3, 6, 10]), seed=seed)
    elif fam == "gridgen": w = max(3, int(n ** 0.5)); inst = generators.gridgen_style(w, w, seed=seed)
    else: w = max(3, int(n ** 0.5)); inst = generators.goto_style(w, w, seed=seed)
    rule = rng.choice([0, 1, 2])
    cold = oracle.emul_solve(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=rule)
    if cold["status"] != "optimal":
        seed += 1
        continue
    it = np.asarray(cold["in_tree"], np.int8)
    au = ((it == 0) & (cold["flow"] == inst.cap) & (inst.cap > 0)).astype(np.int8)
    supply, cost, cap = inst.supply.copy(), inst.cost.copy(), inst.cap.copy()
    kind = rng.choice(["supply", "cost", "cap", "all", "none", "none"])
    nprng = np.random.default_rng(seed)
    if kind in ("supply", "all"):
        a, b = nprng.choice(inst.n, 2, replace=False); k = int(nprng.integers(1, 6)); supply[a] += k; supply[b] -= k
    if kind in ("cost", "all"):
        idx = nprng.choice(inst.m, max(1, inst.m // 50), replace=False); cost[idx] = np.maximum(1, cost[idx] + nprng.integers(-50, 50, idx.size))
    if kind in ("cap", "all"):
        idx = nprng.choice(inst.m, max(1, inst.m // 50), replace=False); cap[idx] = np.maximum(0, cap[idx] + nprng.integers(-20, 20, idx.size))
    with_upper = rng.random() < 0.8   # (without the at-capacity flags the non-basic arcs start at their lower bound: a different point)
    warm = oracle.emul_solve(inst.n, inst.tail, inst.head, cost, cap, supply, rule=rule, warm_in_tree=it, warm_at_upper=au if with_upper else None)
    ref = oracle.emul_solve(inst.n, inst.tail, inst.head, cost, cap, supply, rule=0)
    good = warm["status"] == ref["status"] and (warm["status"] != "optimal" or warm["objective"] == ref["objective"])
    if good and kind == "none" and with_upper and warm["warm_applied"] and inst.n - int(it.sum()) == 1:
        zero += 1
        good = warm["pivots"] == 0
    runs += 1
    applied += bool(warm["warm_applied"])
    if not good:
        bad += 1
        print("MISMATCH", json.dumps({"seed": seed, "family": fam, "n": inst.n, "rule": rule, "kind": kind, "applied": bool(warm["warm_applied"]),
                                      "status": [warm["status"], ref["status"]], "pivots": warm["pivots"]}), flush=True)
    seed += 1
print(json.dumps({"runs": runs, "bad": bad, "basis_applied": applied, "one_component_bases_confirmed_in_zero_pivots": zero}))
sys.exit(1 if bad else 0)
