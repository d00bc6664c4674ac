"""Warm start through the reference-shaped API on the GPU (``-m gpu``).

Scenarios and expected outcomes restate /root/reference/tests/unit/test_warm_start.py (line numbers in
the case table); the data are inputs + expected status / objective, not code.  Outcome parity only: a
warm start changes the pivot count, never the optimum (integer data -> exact comparisons)."""

import logging

import numpy as np
import pytest

import network_flow_solver_amd as nfs
import oracle
from network_flow_solver_amd import generators
from network_flow_solver_amd.data import Basis

pytestmark = pytest.mark.gpu


def _problem(nodes, arcs):
    return nfs.build_problem(nodes=[{"id": i, "supply": s} for i, s in nodes],
                             arcs=[{"tail": t, "head": h, "capacity": c, "cost": w} for t, h, c, w in arcs],
                             directed=True, tolerance=1e-6)


ST = [("s", 10.0), ("t", -10.0)]

# (name, first problem or None, second problem, explicit basis or None, expected status, expected objective)
CASES = [
    ("identical_problem:42-65", _problem(ST, [("s", "t", 20.0, 1.0)]), _problem(ST, [("s", "t", 20.0, 1.0)]), None, "optimal", 10.0),
    ("capacity_increase:67-99", _problem([("s", 100.0), ("t", -100.0)], [("s", "t", 100.0, 1.0)]),
     _problem([("s", 100.0), ("t", -100.0)], [("s", "t", 150.0, 1.0)]), None, "optimal", 100.0),
    ("different_arcs:105-139", _problem([("s", 10.0), ("m", 0.0), ("t", -10.0)], [("s", "m", 20.0, 1.0), ("m", "t", 20.0, 1.0)]),
     _problem(ST, [("s", "t", 20.0, 1.0)]), None, "optimal", 10.0),
    ("empty_basis:141-160", None, _problem(ST, [("s", "t", 20.0, 1.0)]), Basis(tree_arcs=set(), arc_flows={}), "optimal", 10.0),
    ("capacity_decrease_infeasible:162-197",
     _problem([("s", 100.0), ("m", 0.0), ("t", -100.0)], [("s", "m", 100.0, 1.0), ("m", "t", 100.0, 1.0)]),
     _problem([("s", 100.0), ("m", 0.0), ("t", -100.0)], [("s", "m", 50.0, 1.0), ("m", "t", 100.0, 1.0)]), None, "infeasible", 0.0),
    ("supply_change:203-237", _problem([("s", 50.0), ("t", -50.0)], [("s", "t", 100.0, 1.0)]),
     _problem([("s", 75.0), ("t", -75.0)], [("s", "t", 100.0, 1.0)]), None, "optimal", 75.0),
    ("three_components:608-651", None,
     _problem([("n0", 10.0), ("n1", -10.0), ("n2", 8.0), ("n3", -8.0), ("n4", 5.0), ("n5", -5.0)],
              [("n0", "n1", 10.0, 1.0), ("n2", "n3", 8.0, 1.0), ("n4", "n5", 5.0, 1.0), ("n1", "n2", 10.0, 10.0),
               ("n3", "n4", 10.0, 10.0)]),
     Basis(tree_arcs={("n0", "n1"), ("n2", "n3"), ("n4", "n5")},
           arc_flows={("n0", "n1"): 10.0, ("n2", "n3"): 8.0, ("n4", "n5"): 5.0}), "optimal", 23.0),
    ("cost_change:239-275",
     _problem([("s", 10.0), ("m", 0.0), ("t", -10.0)], [("s", "m", 20.0, 1.0), ("m", "t", 20.0, 1.0), ("s", "t", 20.0, 5.0)]),
     _problem([("s", 10.0), ("m", 0.0), ("t", -10.0)], [("s", "m", 20.0, 3.0), ("m", "t", 20.0, 3.0), ("s", "t", 20.0, 1.0)]),
     None, "optimal", 10.0),
    ("add_arc:277-315",
     _problem([("s", 10.0), ("m", 0.0), ("t", -10.0)], [("s", "m", 20.0, 2.0), ("m", "t", 20.0, 2.0)]),
     _problem([("s", 10.0), ("m", 0.0), ("t", -10.0)], [("s", "m", 20.0, 2.0), ("m", "t", 20.0, 2.0), ("s", "t", 20.0, 1.0)]),
     None, "optimal", 10.0),
    ("remove_arc:317-353",
     _problem([("s", 10.0), ("m", 0.0), ("t", -10.0)], [("s", "m", 20.0, 2.0), ("m", "t", 20.0, 2.0), ("s", "t", 20.0, 1.0)]),
     _problem([("s", 10.0), ("m", 0.0), ("t", -10.0)], [("s", "m", 20.0, 2.0), ("m", "t", 20.0, 2.0)]),
     None, "optimal", 40.0),
    ("multi_commodity_style:427-459",
     _problem([("s1", 30.0), ("s2", 40.0), ("m", 0.0), ("t1", -35.0), ("t2", -35.0)],
              [("s1", "m", 50.0, 1.0), ("s2", "m", 50.0, 1.0), ("m", "t1", 50.0, 1.0), ("m", "t2", 50.0, 1.0)]),
     _problem([("s1", 30.0), ("s2", 40.0), ("m", 0.0), ("t1", -35.0), ("t2", -35.0)],
              [("s1", "m", 60.0, 1.0), ("s2", "m", 50.0, 1.0), ("m", "t1", 50.0, 1.0), ("m", "t2", 50.0, 1.0)]),
     None, "optimal", 140.0),
    ("with_cycles:461-492",
     _problem([("s", 10.0), ("a", 0.0), ("b", 0.0), ("t", -10.0)],
              [("s", "a", 20.0, 1.0), ("a", "b", 20.0, 1.0), ("b", "a", 20.0, 2.0), ("a", "t", 20.0, 1.0)]),
     _problem([("s", 10.0), ("a", 0.0), ("b", 0.0), ("t", -10.0)],
              [("s", "a", 20.0, 2.0), ("a", "b", 20.0, 1.0), ("b", "a", 20.0, 2.0), ("a", "t", 20.0, 2.0)]),
     None, "optimal", 40.0),
    ("disconnected_components:556-606", None,
     _problem([("s1", 20.0), ("t1", -20.0), ("s2", 15.0), ("t2", -15.0), ("m1", 0.0), ("m2", 0.0)],
              [("s1", "m1", 20.0, 1.0), ("m1", "t1", 20.0, 1.0), ("s2", "m2", 15.0, 1.0), ("m2", "t2", 15.0, 1.0),
               ("m1", "m2", 10.0, 5.0)]),
     Basis(tree_arcs={("s1", "m1"), ("m1", "t1")}, arc_flows={("s1", "m1"): 20.0, ("m1", "t1"): 20.0}), "optimal", 70.0),
    ("single_arc_basis:653-686", None,
     _problem([("a", 15.0), ("b", 0.0), ("c", 0.0), ("d", -15.0)],
              [("a", "b", 15.0, 1.0), ("b", "c", 15.0, 1.0), ("c", "d", 15.0, 1.0), ("a", "d", 10.0, 4.0)]),
     Basis(tree_arcs={("a", "b")}, arc_flows={("a", "b"): 15.0}), "optimal", 45.0),
]


@pytest.mark.parametrize("name,first,second,basis,status,objective", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("strategy", ["dantzig", "devex", "candidate_list"])
def test_reference_warm_start_scenarios(gpu_engine_module, name, first, second, basis, status, objective, strategy):
    opts = nfs.SolverOptions(pricing_strategy=strategy, explicit_pricing_strategy=True)
    if first is not None:
        r1 = nfs.solve_min_cost_flow(first, options=opts)
        assert r1.status == "optimal" and r1.basis is not None
        basis = r1.basis
    r2 = nfs.solve_min_cost_flow(second, options=opts, warm_start_basis=basis)
    assert r2.status == status
    assert r2.objective == objective
    cold = nfs.solve_min_cost_flow(second, options=opts)
    assert (cold.status, cold.objective, cold.flows) == (r2.status, r2.objective, r2.flows)
    if name.startswith("identical"):
        assert r2.iterations <= r1.iterations


def test_none_basis_is_a_cold_start(gpu_engine_module):
    """test_warm_start.py:22-40."""
    r = nfs.solve_min_cost_flow(_problem(ST, [("s", "t", 20.0, 1.0)]), warm_start_basis=None)
    assert r.status == "optimal" and r.objective == 10.0 and r.basis is not None


def test_sequential_warm_starts_with_growing_capacity(gpu_engine_module):
    """test_warm_start.py:359-390: the basis of each solve starts the next; the optimum does not move."""
    basis, prev = None, None
    for capacity in (100.0, 110.0, 120.0, 130.0):
        r = nfs.solve_min_cost_flow(_problem([("s", 100.0), ("t", -100.0)], [("s", "t", capacity, 1.0)]), warm_start_basis=basis)
        assert r.status == "optimal"
        if prev is not None:
            assert r.objective == prev
        basis, prev = r.basis, r.objective


def test_warm_start_that_needs_no_artificial_arc_logs_it(gpu_engine_module, caplog):
    """test_warm_start.py:755-790 (the reference logs the Phase-1 skip; here: the basis is applied and the solve only
    confirms optimality)."""
    p = _problem([("s", 25.0), ("m", 0.0), ("t", -25.0)], [("s", "m", 30.0, 1.0), ("m", "t", 30.0, 1.0)])
    r1 = nfs.solve_min_cost_flow(p)
    with caplog.at_level(logging.INFO):
        r2 = nfs.solve_min_cost_flow(p, warm_start_basis=r1.basis)
    assert r2.status == "optimal" and r2.objective == r1.objective == 50.0 and r2.iterations <= r1.iterations
    msgs = [rec.getMessage().lower() for rec in caplog.records]
    assert msgs and any("warm" in m or "phase" in m for m in msgs)


def test_sequential_warm_starts_with_increasing_demand(gpu_engine_module):
    """test_warm_start.py:392-424."""
    basis, prev = None, 0.0
    for demand in (50.0, 75.0, 100.0, 125.0, 150.0):
        p = _problem([("s", demand), ("t", -demand)], [("s", "t", 200.0, 1.0)])
        r = nfs.solve_min_cost_flow(p, warm_start_basis=basis)
        assert r.status == "optimal" and r.objective == demand and r.objective >= prev
        basis, prev = r.basis, r.objective


def test_basis_extraction(gpu_engine_module):
    """test_warm_start.py:498-550: tree arcs carry flows; basis flows equal result flows."""
    p = _problem([("s", 10.0), ("m", 0.0), ("t", -10.0)], [("s", "m", 20.0, 1.0), ("m", "t", 20.0, 1.0), ("s", "t", 20.0, 5.0)])
    r = nfs.solve_min_cost_flow(p)
    assert r.status == "optimal" and isinstance(r.basis.tree_arcs, set) and isinstance(r.basis.arc_flows, dict)
    assert 1 <= len(r.basis.tree_arcs) <= 3
    for arc in r.basis.tree_arcs:
        assert arc in r.basis.arc_flows
        if arc in r.flows:
            assert r.basis.arc_flows[arc] == r.flows[arc]


def test_warm_start_logging(gpu_engine_module, caplog):
    """test_warm_start.py:692-753: the attempt, the success and the fall-back are logged."""
    p = _problem(ST, [("s", "t", 20.0, 1.0)])
    r1 = nfs.solve_min_cost_flow(p)
    with caplog.at_level(logging.INFO):
        nfs.solve_min_cost_flow(p, warm_start_basis=r1.basis)
        nfs.solve_min_cost_flow(p, warm_start_basis=Basis(tree_arcs=set(), arc_flows={}))
    text = " ".join(rec.getMessage() for rec in caplog.records)
    assert "Attempting to apply warm-start basis" in text and "empty" in text.lower()


@pytest.mark.parametrize("rule", [0, 1, 2], ids=["dantzig", "devex_block", "candidate_list"])
@pytest.mark.parametrize("name", ["netgen_8_08a", "netgen_8_10a", "gridgen_8_14a"])
def test_engine_warm_start_at_scale(gpu_engine_module, name, rule):
    """Raw C ABI (mcf_set_basis) on all three engine paths (LDS loop, persistent loop, kernel per phase):
    the optimal basis re-installed needs only degenerate pivots; after a supply change the warm solve reaches
    the cold optimum (checked against the CPU emulation's cold solve) in fewer pivots; a rejected basis leaves
    the handle at the cold start; mcf_reset returns to the all-artificial basis."""
    e = gpu_engine_module
    inst = generators.named_instance(name)
    with e.McfEngine(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=rule) as eng:
        eng.solve()
        cold = eng.result()
        in_tree = cold.in_tree.astype(np.int8)
        at_upper = (~cold.in_tree & (cold.flow == inst.cap) & (inst.cap > 0)).astype(np.int8)
        assert eng.set_basis(in_tree, at_upper)
        eng.solve()
        warm = eng.result()
        assert warm.status == "optimal" and warm.objective == cold.objective and np.array_equal(warm.flow, cold.flow)
        assert warm.stats["pivots"] == warm.stats["degenerate"] <= max(3, cold.stats["pivots"] // 20)
        assert not eng.set_basis(np.ones(inst.m, np.int8))          # cycles: rejected ...
        eng.solve()
        again = eng.result()                                         # ... and solved from the cold start
        assert again.objective == cold.objective and again.stats["pivots"] == cold.stats["pivots"]
        eng.set_basis(in_tree, at_upper)
        eng.reset()
        eng.solve()
        assert eng.result().stats["pivots"] == cold.stats["pivots"]
    supply = inst.supply.copy()
    supply[int(np.argmax(supply))] += 5
    supply[int(np.argmin(supply))] -= 5
    ref = oracle.emul_solve(inst.n, inst.tail, inst.head, inst.cost, inst.cap, supply, rule=0)
    with e.McfEngine(inst.n, inst.tail, inst.head, inst.cost, inst.cap, supply, rule=rule) as eng:
        applied = eng.set_basis(in_tree, at_upper)
        eng.solve()
        r = eng.result()
        assert r.status == ref["status"] == "optimal" and r.objective == ref["objective"]
        if applied:
            assert r.stats["pivots"] < cold.stats["pivots"] // 2
