#!/usr/bin/env python3
"""How many pivots a warm start from a certified optimal basis takes, against the number of artificial arcs the optimal tree
keeps (the forest components the Basis hands over).  python scripts/warm_start_pivots.py [instance ...]"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import network_flow_solver_amd as nfs  # noqa: E402
from network_flow_solver_amd import engine, generators  # noqa: E402
from network_flow_solver_amd.data import ArrayBasis  # noqa: E402

for name in sys.argv[1:] or ["netgen_8_14a", "gridgen_8_13a", "goto_8_14a"]:
    inst = generators.named_instance(name)
    with engine.McfEngine(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=2) as eng:
        eng.solve(max_pivots=100_000_000)
        res = eng.result()
    k = int(inst.n - int(res.in_tree.sum()))
    at_upper = ~res.in_tree & (inst.cap > 0) & (res.flow == inst.cap)
    zero_tree = int((res.in_tree & (res.flow == 0)).sum())
    full_tree = int((res.in_tree & (inst.cap > 0) & (res.flow == inst.cap)).sum())
    prob = nfs.SoAProblem(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply)
    api = nfs.solve_min_cost_flow(prob, warm_start_basis=ArrayBasis(inst.tail, inst.head, res.in_tree, at_upper, res.flow))
    rc = inst.cost + res.potential[inst.tail] - res.potential[inst.head]
    tied = int((~res.in_tree & (rc == 0)).sum())
    print(f"{name}: cold {res.stats['pivots']} pivots; artificial basic arcs k={k}; degenerate tree arcs at 0: {zero_tree}, at cap: {full_tree}; "
          f"non-tree arcs with zero reduced cost: {tied}; warm start: {api.status} in {api.iterations} pivots, same flow: {np.array_equal(api.flows.array, res.flow)}, "
          f"objective equal: {api.objective == float(res.objective)}", flush=True)
