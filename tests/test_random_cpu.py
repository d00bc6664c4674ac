"""Randomised outcome parity on the CPU: the oracle (reference algorithm restated) against the
engine's own integer algorithm (CPU emulation) fed through the product's `flatten_problem`
(ordering, lower-bound shift, decimal scaling).  Exercises negative costs, zero / unlimited
capacities, lower bounds, parallel arcs, undirected edges, fractional data, infeasibility."""

import numpy as np
import pytest

import network_flow_solver_amd as nfs
import oracle
from network_flow_solver_amd.simplex import flatten_problem
from random_instances import make

SEEDS = list(range(160))


def engine_outcome_via_emulation(problem, rule):
    f = flatten_problem(problem)
    r = oracle.emul_solve(len(f.node_ids), f.tail, f.head, f.cost, f.cap, f.supply, rule=rule)
    if r["status"] in ("infeasible", "unbounded"):
        return r["status"], None
    flow = r["flow"].astype(np.float64) / f.flow_scale + f.lower
    return r["status"], float(np.dot(flow, f.orig_cost))


def networkx_truth(problem):
    """Independent exact solve of the flattened integer instance with networkx."""
    nx = pytest.importorskip("networkx")
    f = flatten_problem(problem)
    g = nx.MultiDiGraph()
    for i in range(len(f.node_ids)):
        g.add_node(i, demand=-int(f.supply[i]))
    for t, h, c, cap in zip(f.tail, f.head, f.cost, f.cap):
        if cap < 0:
            g.add_edge(int(t), int(h), weight=int(c))
        else:
            g.add_edge(int(t), int(h), weight=int(c), capacity=int(cap))
    try:
        cost, _ = nx.network_simplex(g)
    except nx.NetworkXUnfeasible:
        return "infeasible", None
    except nx.NetworkXUnbounded:
        return "unbounded", None
    return "optimal", cost / (f.flow_scale * f.cost_scale) + float(np.dot(f.lower, f.orig_cost))


@pytest.mark.parametrize("seed", SEEDS)
def test_random_problem_outcomes(seed):
    """Engine algorithm == ground truth (networkx) on every instance; the oracle (= the reference's
    behaviour) == ground truth wherever the reference's own two strategies agree.  Where they
    disagree the reference is defective on one side -- its vectorised Devex prices Phase 1 with
    stale costs (seed 12: devex says "infeasible", truth -412.0) and
    its Dantzig loop can cycle to the iteration limit on an infeasible input (seed 143) -- and the
    oracle reproduces either behaviour pivot for pivot (checked against the reference itself in the
    build container); those are not parity targets."""
    nodes, arcs, directed = make(seed)
    problem = nfs.build_problem(nodes, arcs, directed, 1e-6)
    truth_status, truth_obj = networkx_truth(problem)
    for rule in (0, 1):
        status, objective = engine_outcome_via_emulation(problem, rule)
        assert status == truth_status, (seed, rule, status, truth_status)
        if status == "optimal":
            assert objective == pytest.approx(truth_obj, abs=1e-7)
    d = oracle.solve_dicts(nodes, arcs, directed, 1e-6, "dantzig")
    x = oracle.solve_dicts(nodes, arcs, directed, 1e-6, "devex")
    if d.status == x.status:
        assert d.status == truth_status
        if d.status == "optimal":
            assert d.objective == pytest.approx(truth_obj, abs=1e-7) and x.objective == pytest.approx(truth_obj, abs=1e-7)
    else:
        assert truth_status in (d.status, x.status)


def test_reference_defects_are_reproduced_not_copied():
    """Seeds 12 and 143: the reference's own two strategies disagree (verified against the reference
    in the build container).  The oracle reproduces both sides; the engine returns the truth."""
    nodes, arcs, directed = make(12)
    d = oracle.solve_dicts(nodes, arcs, directed, 1e-6, "dantzig")
    x = oracle.solve_dicts(nodes, arcs, directed, 1e-6, "devex")
    assert (d.status, d.objective, d.iterations) == ("optimal", -412.0, 7)
    assert (x.status, x.iterations) == ("infeasible", 9)
    nodes, arcs, directed = make(143)
    d = oracle.solve_dicts(nodes, arcs, directed, 1e-6, "dantzig")
    x = oracle.solve_dicts(nodes, arcs, directed, 1e-6, "devex")
    assert (d.status, d.iterations) == ("iteration_limit", 160) and (x.status, x.iterations) == ("infeasible", 1)
