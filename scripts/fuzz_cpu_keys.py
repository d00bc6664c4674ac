#!/usr/bin/env python3
"""CPU-only fuzz: the emulation under every key variant (plain, forward first, priority bits, capacity merit), Dantzig and
candidate list, against the optimum of the pinned oracle (oracle/ref_simplex.c) + the optimality certificate.
    python scripts/fuzz_cpu_keys.py [seconds]"""
import sys, time, random
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, oracle
from network_flow_solver_amd import generators
from conftest import check_optimality
t_end = time.time() + (float(sys.argv[1]) if len(sys.argv) > 1 else 240)
seed = 9000; runs = 0; bad = 0
while time.time() < t_end:
    rng = random.Random(seed)
    n = rng.choice([20, 50, 120, 300, 700])
    fam = rng.choice(["netgen", "gridgen", "goto"])
    if fam == "netgen": inst = generators.netgen_style(n, n * rng.choice([3, 6, 10]), seed=seed)
    elif fam == "gridgen": w = max(4, int(n ** 0.5)); inst = generators.gridgen_style(w, w, seed=seed)
    else: w = max(4, int(n ** 0.5)); inst = generators.goto_style(w, w, seed=seed)
    ref = oracle.solve_soa(inst, "dantzig", reference_order=False)
    for rule in (0, 2):
        for km in (0, 1, 2, 3):
            prio = np.random.default_rng(seed).integers(0, 4, size=inst.m).astype(np.int8) if km == 2 else None
            em = oracle.emul_solve(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=rule | (km << 8), arc_priority=prio,
                                   climb_budget=rng.choice([-1, 0, 2]))
            ok = em["status"] == ref["status"] and (em["status"] != "optimal" or em["objective"] == int(round(ref["objective"])))
            if ok and em["status"] == "optimal":
                try: check_optimality(inst, em["flow"], em["potential"])
                except AssertionError: ok = False
            runs += 1
            if not ok:
                bad += 1; print("MISMATCH", seed, fam, n, rule, km, em["status"], em["objective"], ref["status"], ref["objective"], flush=True)
    seed += 1
print({"instances": seed - 9000, "runs": runs, "bad": bad})
