#!/usr/bin/env python3
"""Golden vectors for the reference's specialised pivot strategies on NON-trivial structured instances (its own fixtures
of these classes are an unbounded and an infeasible toy): seeded shortest-path, max-flow and bipartite-matching problems,
solved by the reference itself.  Runs ONLY in the build container (imports /root/reference/src).

    python3 tests/golden/make_structured.py        ->  tests/golden/structured_cases.json
"""
import json
import logging
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, "/root/reference/src")
from network_solver import SolverOptions, build_problem, solve_min_cost_flow  # type: ignore  # noqa: E402
from network_solver.specializations import analyze_network_structure  # type: ignore  # noqa: E402

logging.disable(logging.CRITICAL)


def structured_instances():
    """Seeded instances of the three classes the reference has heuristics for beyond transportation / assignment."""
    rng = np.random.default_rng(5)
    out = []
    # shortest path: one unit from s to t over a layered graph with shortcuts
    layers, width = 6, 5
    name = lambda l, k: f"n{l}_{k}"
    nodes = [{"id": "s", "supply": 1.0}, {"id": "t", "supply": -1.0}] + [{"id": name(l, k), "supply": 0.0} for l in range(layers) for k in range(width)]
    arcs = [{"tail": "s", "head": name(0, k), "cost": float(rng.integers(1, 9)), "capacity": 3.0} for k in range(width)]
    arcs += [{"tail": name(layers - 1, k), "head": "t", "cost": float(rng.integers(1, 9)), "capacity": 3.0} for k in range(width)]
    for l in range(layers - 1):
        for k in range(width):
            for k2 in rng.choice(width, size=3, replace=False):
                arcs.append({"tail": name(l, k), "head": name(l + 1, int(k2)), "cost": float(rng.integers(1, 20)), "capacity": 2.0})
    arcs.append({"tail": "orphan", "head": name(2, 0), "cost": 1.0, "capacity": 1.0})     # a tail the source does not reach
    nodes.append({"id": "orphan", "supply": 0.0})
    out.append(("shortest_path", nodes, arcs))
    # max flow in the reference's sense: one source, one sink, several units, uniform costs
    nodes = [{"id": "s", "supply": 7.0}, {"id": "t", "supply": -7.0}] + [{"id": name(l, k), "supply": 0.0} for l in range(3) for k in range(4)]
    arcs = [{"tail": "s", "head": name(0, k), "cost": 1.0, "capacity": float(rng.integers(2, 6))} for k in range(4)]
    arcs += [{"tail": name(2, k), "head": "t", "cost": 1.0, "capacity": float(rng.integers(2, 6))} for k in range(4)]
    for l in range(2):
        for k in range(4):
            for k2 in range(4):
                arcs.append({"tail": name(l, k), "head": name(l + 1, k2), "cost": 1.0, "capacity": float(rng.integers(1, 4)) if (k + k2) % 3 else None})
    out.append(("max_flow", nodes, arcs))
    # bipartite matching: unit supplies / demands plus idle nodes on both sides
    k = 9
    nodes = [{"id": f"l{i}", "supply": 1.0 if i < k else 0.0} for i in range(k + 3)] + [{"id": f"r{i}", "supply": -1.0 if i < k else 0.0} for i in range(k + 3)]
    arcs = [{"tail": f"l{i}", "head": f"r{i}", "cost": float(rng.integers(5, 30)), "capacity": 1.0} for i in range(k + 3)]
    for i in range(k + 3):
        for j in rng.choice(k + 3, size=4, replace=False):
            if int(j) != i:
                arcs.append({"tail": f"l{i}", "head": f"r{int(j)}", "cost": float(rng.integers(1, 30)), "capacity": 1.0})
    out.append(("bipartite_matching", nodes, arcs))
    return out



cases = []
for kind, nodes, arcs in structured_instances():
    problem = build_problem(nodes=nodes, arcs=arcs, directed=True, tolerance=1e-6)
    assert analyze_network_structure(problem).network_type.value == kind
    expected = {}
    for strategy in ("dantzig", "devex"):
        res = solve_min_cost_flow(problem, options=SolverOptions(auto_scale=False, pricing_strategy=strategy, explicit_pricing_strategy=True),
                                  max_iterations=2000)
        expected[strategy] = {"status": res.status, "objective": float(res.objective), "iterations": int(res.iterations),
                              "flows": [[t, h, float(f)] for (t, h), f in sorted(res.flows.items())]}
        print(kind, strategy, res.status, res.objective, res.iterations)
    cases.append({"name": f"structured_{kind}", "network_type": kind, "directed": True, "tolerance": 1e-6, "max_iterations": 2000,
                  "nodes": nodes, "arcs": arcs, "expected": expected})
(HERE / "structured_cases.json").write_text(json.dumps(cases, indent=1) + "\n")
