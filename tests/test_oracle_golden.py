"""The oracle (oracle/ref_simplex.c, a C restatement of the reference's algorithm) against
every golden vector produced by the reference itself (tests/golden/make_golden.py).
This is what pins the oracle; the GPU parity tests then lean on the oracle."""

import pytest

import oracle
from conftest import CASE_IDS, CASES, golden_flows, load_synthetic

STRATS = ("dantzig", "devex")

# transportation / assignment / bipartite-matching / max-flow / shortest-path inputs make the reference try a
# specialised pivot strategy first (simplex.py:1061-1064, specialized_pivots.py:69-527); the oracle restates the
# structure analysis (specializations.py:60-288) and all five strategies, so pivot counts are pinned on EVERY case.


@pytest.mark.parametrize("case", CASES, ids=CASE_IDS)
@pytest.mark.parametrize("strategy", STRATS)
def test_small_cases_match_reference(case, strategy):
    exp = case["expected"][strategy]
    res = oracle.solve_dicts(case["nodes"], case["arcs"], case["directed"], case["tolerance"], strategy,
                             max_iterations=case.get("max_iterations"))
    assert res.status == exp["status"]
    if exp["status"] == "unbounded":
        return
    assert res.objective == pytest.approx(exp["objective"], abs=1e-9)
    if res.min_nonbasic_abs_rc > 1e-6:
        # dual non-degenerate => the optimal flow is unique => it must be the reference's
        assert res.flows == golden_flows(exp)
    else:
        # alternative optima exist (e.g. tiny_transportation has two flows of cost 111):
        # any of them is acceptable, but it must be feasible and cost the same
        supplies = {str(nd["id"]): float(nd.get("supply", 0.0)) for nd in case["nodes"]}
        for (t, h), f in res.flows.items():
            supplies[t] -= f
            supplies[h] += f
        assert all(abs(v) <= 1e-6 for v in supplies.values())


@pytest.mark.parametrize("case", CASES, ids=CASE_IDS)
def test_network_type_detection_matches_reference(case):
    """The restated structure analysis classifies every fixture the way the reference does (``network_type`` was
    recorded by tests/golden/make_golden.py from analyze_network_structure)."""
    kind, left = oracle.detect_network_type(case["nodes"], case["arcs"], case["directed"], case["tolerance"])
    assert kind == case["network_type"]
    if kind == "bipartite_matching":
        assert left


@pytest.mark.parametrize("case", CASES, ids=CASE_IDS)
@pytest.mark.parametrize("strategy", STRATS)
def test_pivot_counts_match_reference_on_every_case(case, strategy):
    """Pivot for pivot: the iteration count of the reference itself, for the Dantzig loop and the vectorised Devex
    path, on general networks AND on the 12 fixtures where a specialised pivot strategy runs first."""
    exp = case["expected"][strategy]
    if exp.get("iterations") is None:          # unbounded: the reference raises, no count recorded
        return
    res = oracle.solve_dicts(case["nodes"], case["arcs"], case["directed"], case["tolerance"], strategy,
                             max_iterations=case.get("max_iterations"))
    assert res.iterations == exp["iterations"]


def _structured_cases():
    import json
    from pathlib import Path
    return json.loads((Path(__file__).parent / "golden" / "structured_cases.json").read_text())


@pytest.mark.parametrize("case", _structured_cases(), ids=lambda c: c["name"])
@pytest.mark.parametrize("strategy", STRATS)
def test_specialised_strategies_on_real_structured_instances(case, strategy):
    """Shortest-path, max-flow and bipartite-matching instances of some size (tests/golden/make_structured.py ran the
    reference on them): the oracle classifies them like the reference and reproduces its status, objective, pivot count
    and flows -- including the bipartite-matching heuristic's failure to terminate (iteration limit after 2 000 pivots)."""
    exp = case["expected"][strategy]
    assert oracle.detect_network_type(case["nodes"], case["arcs"], True, case["tolerance"])[0] == case["network_type"]
    res = oracle.solve_dicts(case["nodes"], case["arcs"], True, case["tolerance"], strategy, max_iterations=case["max_iterations"])
    assert (res.status, res.iterations) == (exp["status"], exp["iterations"])
    assert res.objective == pytest.approx(exp["objective"], abs=1e-9)
    if res.status == "optimal" and res.min_nonbasic_abs_rc > 1e-6:
        assert res.flows == pytest.approx({(t, h): f for t, h, f in exp["flows"] if abs(f) > 1e-9})


def test_specialised_strategies_change_the_pivot_sequence():
    """The specialised rules are not a no-op: switched off (special=0), at least one transportation fixture takes a
    different number of pivots than the reference did."""
    differs = 0
    for case in CASES:
        if case["network_type"] in ("transportation", "assignment"):
            for strategy in STRATS:
                plain = oracle.solve_dicts(case["nodes"], case["arcs"], case["directed"], case["tolerance"], strategy,
                                           special=0)
                assert plain.objective == pytest.approx(case["expected"][strategy]["objective"], abs=1e-9)
                differs += plain.iterations != case["expected"][strategy]["iterations"]
    assert differs > 0


@pytest.mark.parametrize("entry,inst", load_synthetic(), ids=lambda x: x["name"] if isinstance(x, dict) else "")
def test_synthetic_instances_pivot_for_pivot(entry, inst):
    """GENERAL networks: same status, objective, flows AND iteration count as the reference,
    for both the Dantzig loop and the vectorised Devex path."""
    for strategy, exp in entry["expected"].items():
        res = oracle.solve_soa(inst, strategy)
        assert res["status"] == exp["status"] == "optimal"
        assert res["objective"] == pytest.approx(exp["objective"], abs=1e-6)
        assert res["iterations"] == exp["iterations"], strategy
        got = {(int(inst.tail[i]), int(inst.head[i])): float(res["flow"][i]) for i in range(inst.m)
               if abs(res["flow"][i]) > 1e-6}
        assert got == golden_flows(exp)


def test_candidate_list_and_adaptive_reach_the_same_optimum():
    """simplex_pricing.py:375-639 restated; the reference's own cross-strategy check is
    tests/unit/test_pricing_strategies.py:250-294."""
    entry, inst = load_synthetic()[0]
    want = entry["expected"]["dantzig"]["objective"]
    for strategy in ("candidate_list", "adaptive"):
        res = oracle.solve_soa(inst, strategy)
        assert res["status"] == "optimal" and res["objective"] == pytest.approx(want, abs=1e-6)


def test_loop_devex_matches_vectorised_objective():
    entry, inst = load_synthetic()[0]
    res = oracle.solve_soa(inst, "devex", use_vectorized_pricing=False)
    assert res["objective"] == pytest.approx(entry["expected"]["devex"]["objective"], abs=1e-6)
