"""CPU emulation of the engine's own integer pivot algorithm (same headers as the HIP
kernels, scalar loops) against the reference-derived goldens, plus structural invariants
of the preorder-array spanning tree.  No GPU needed."""

import numpy as np
import pytest

import oracle
from conftest import check_optimality, check_tree_invariants, golden_flows, load_synthetic, optimum_is_unique


@pytest.mark.parametrize("entry,inst", load_synthetic(), ids=lambda x: x["name"] if isinstance(x, dict) else "")
@pytest.mark.parametrize("rule", [0, 1, 2], ids=["dantzig", "devex_block", "candidate_list"])
def test_integer_engine_reaches_reference_optimum(entry, inst, rule):
    exp = next(iter(entry["expected"].values()))
    res = oracle.emul_solve(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=rule)
    assert res["status"] == "optimal"
    assert res["objective"] == int(round(exp["objective"]))          # bit-exact integer objective
    check_tree_invariants(inst.n, res["parent"], res["size"], res["pos"], res["order"], res["depth"], res["psize"])
    rc = check_optimality(inst, res["flow"], res["potential"])
    if optimum_is_unique(inst, res["flow"], res["in_tree"], rc):
        got = {(int(inst.tail[i]), int(inst.head[i])): float(res["flow"][i]) for i in range(inst.m) if res["flow"][i]}
        assert got == golden_flows(exp)


def test_tree_invariants_hold_after_every_pivot():
    _, inst = load_synthetic()[0]
    for cap in range(1, 140):
        res = oracle.emul_solve(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=0, max_pivots=cap)
        check_tree_invariants(inst.n, res["parent"], res["size"], res["pos"], res["order"], res["depth"], res["psize"])
        if res["status"] == "optimal":
            break
    assert res["status"] == "optimal"


def test_infeasible_and_unbounded_verdicts():
    # capacity-starved: tests/integration/test_unbounded_detection.py:37-55
    r = oracle.emul_solve(3, [0], [1], [1], [5], [5, 0, -5])
    assert r["status"] == "infeasible" and r["artificial_flow"] > 0
    # negative cycle without capacity: test_unbounded_detection.py:15-34
    r = oracle.emul_solve(2, [0, 1], [1, 0], [-5, 1], [-1, -1], [0, 0])
    assert r["status"] == "unbounded" and r["unbounded_arc"] in (0, 1)


def test_zero_capacity_and_empty_arc_set():
    r = oracle.emul_solve(2, [0], [1], [3], [0], [0, 0])
    assert r["status"] == "optimal" and r["objective"] == 0
    r = oracle.emul_solve(2, np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.int64),
                          np.zeros(0, np.int64), [0, 0])
    assert r["status"] == "optimal" and r["objective"] == 0
    r = oracle.emul_solve(2, np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.int64),
                          np.zeros(0, np.int64), [4, -4])
    assert r["status"] == "infeasible"


def test_pivot_budget_reports_iteration_limit():
    _, inst = load_synthetic()[0]
    r = oracle.emul_solve(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=0, max_pivots=5)
    assert r["status"] == "iteration_limit" and r["pivots"] == 5


def test_candidate_list_sweeps_far_less_than_dantzig():
    """simplex_pricing.py:375-542 in engine form: same optimum, about the same number of pivots,
    a fraction of the full sweeps (most pivots come from re-pricing the list)."""
    _, inst = load_synthetic()[7]
    d = oracle.emul_solve(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=0)
    c = oracle.emul_solve(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=2)
    assert c["objective"] == d["objective"] and c["status"] == "optimal"
    assert c["major_sweeps"] + c["minor_pivots"] >= c["pivots"] and c["minor_pivots"] > c["pivots"] // 2
    assert c["major_sweeps"] < d["pivots"] // 2 and c["arcs_priced"] < d["arcs_priced"] // 2
    assert c["pivots"] < 1.3 * d["pivots"]


@pytest.mark.parametrize("entry,inst", load_synthetic(), ids=lambda x: x["name"] if isinstance(x, dict) else "")
@pytest.mark.parametrize("budget", [0, 3], ids=["scan_only", "climb3_then_scan"])
def test_cycle_scan_reproduces_the_climb(entry, inst, budget):
    """mcf_pivot_scan (cycle found by a sweep over preorder positions with the position-space subtree
    sizes) must hand the ratio test exactly what the pointer-chasing climb hands it: same entering arcs,
    same leaving arcs, hence the same pivot sequence, flows, potentials and tree."""
    kw = dict(rule=0, trace=100000)
    a = oracle.emul_solve(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, climb_budget=-1, **kw)
    b = oracle.emul_solve(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, climb_budget=budget, **kw)
    assert a["scans"] == 0 and b["scans"] > 0
    assert b["pivots"] == a["pivots"] and np.array_equal(a["trace"], b["trace"])
    for key in ("flow", "potential", "parent", "pred_arc", "size", "pos", "order", "depth", "psize"):
        assert np.array_equal(a[key], b[key]), key
    assert b["degenerate"] == a["degenerate"] and b["cycle_arcs"] == a["cycle_arcs"]
    check_tree_invariants(inst.n, b["parent"], b["size"], b["pos"], b["order"], b["depth"], b["psize"])


def test_cycle_scan_on_devex_and_candidate_list_rules():
    _, inst = load_synthetic()[3]
    for rule in (1, 2):
        a = oracle.emul_solve(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=rule, trace=100000)
        b = oracle.emul_solve(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=rule, trace=100000,
                              climb_budget=0)
        assert np.array_equal(a["trace"], b["trace"]) and np.array_equal(a["flow"], b["flow"])
        assert b["objective"] == a["objective"] and b["scans"] > 0


# ------------------------------------------------------------------ warm start (mcf_apply_basis, shared host code)
def _basis_of(inst, res):
    in_tree = np.asarray(res["in_tree"], dtype=np.int8)
    at_upper = ((in_tree == 0) & (res["flow"] == inst.cap) & (inst.cap > 0)).astype(np.int8)
    return in_tree, at_upper


@pytest.mark.parametrize("idx", [0, 1, 2, 3, 5, 6, 7])
def test_warm_start_from_the_optimal_basis_needs_almost_no_pivots(idx):
    """simplex.py:740-1010: re-installing the final basis of a solve (tree arcs + which non-basic arcs sit at
    capacity) reproduces the optimal flows by conservation, and -- since [r3] mcf_apply_basis hangs each component of the
    basis where the fewest degenerate arcs point the wrong way (none, for a basis taken from a strongly feasible tree) --
    the optimum is CONFIRMED in zero pivots when the real basic arcs span the nodes (k == 1 artificial arc left in the tree);
    with k > 1 components the potentials between components may shift and a few degenerate pivots put that right."""
    _, inst = load_synthetic()[idx]
    cold = oracle.emul_solve(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply)
    it, au = _basis_of(inst, cold)
    warm = oracle.emul_solve(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, warm_in_tree=it, warm_at_upper=au)
    assert warm["warm_applied"] and warm["status"] == "optimal" and warm["objective"] == cold["objective"]
    k = inst.n - int(it.sum())
    assert warm["pivots"] == 0 if k == 1 else warm["pivots"] <= max(3, cold["pivots"] // 20), (k, warm["pivots"])
    assert warm["pivots"] == warm["degenerate"]  # the flow was already optimal
    assert np.array_equal(warm["flow"], cold["flow"])
    check_tree_invariants(inst.n, warm["parent"], warm["size"], warm["pos"], warm["order"], warm["depth"], warm["psize"])
    check_optimality(inst, warm["flow"], warm["potential"])


def test_warm_start_after_a_supply_change_reaches_the_cold_optimum():
    _, inst = load_synthetic()[3]
    cold0 = oracle.emul_solve(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply)
    it, au = _basis_of(inst, cold0)
    supply = inst.supply.copy()
    src, dst = int(np.argmax(supply)), int(np.argmin(supply))
    supply[src] += 7
    supply[dst] -= 7
    cold = oracle.emul_solve(inst.n, inst.tail, inst.head, inst.cost, inst.cap, supply)
    warm = oracle.emul_solve(inst.n, inst.tail, inst.head, inst.cost, inst.cap, supply, warm_in_tree=it, warm_at_upper=au)
    assert warm["status"] == cold["status"] == "optimal" and warm["objective"] == cold["objective"]
    if warm["warm_applied"]:
        assert warm["pivots"] < cold["pivots"]


def test_warm_start_rejects_cycles_and_empty_bases_then_solves_cold():
    _, inst = load_synthetic()[0]
    cold = oracle.emul_solve(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply)
    everything = np.ones(inst.m, np.int8)                          # 512 arcs on 64 nodes: cycles
    r = oracle.emul_solve(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, warm_in_tree=everything)
    assert not r["warm_applied"] and r["objective"] == cold["objective"] and r["pivots"] == cold["pivots"]
    r = oracle.emul_solve(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, warm_in_tree=np.zeros(inst.m, np.int8))
    assert not r["warm_applied"] and r["objective"] == cold["objective"]


def test_warm_start_without_bound_information_still_reaches_the_optimum():
    """What the reference's Basis carries: tree arcs only, non-basic arcs assumed at zero."""
    for idx in (0, 3, 6):
        _, inst = load_synthetic()[idx]
        cold = oracle.emul_solve(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply)
        it, _ = _basis_of(inst, cold)
        r = oracle.emul_solve(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, warm_in_tree=it)
        assert r["status"] == "optimal" and r["objective"] == cold["objective"]
        check_optimality(inst, r["flow"], r["potential"])


def test_cycle_scan_above_65535_nodes():
    """Scan vs climb, pivot for pivot, on 70 000 nodes (several scan rounds per pivot on the device; a halfword
    variant of the position-space sizes was measured and dropped -- this size is where it would saturate)."""
    from network_flow_solver_amd import generators
    inst = generators.netgen_style(70000, 280000, seed=3)
    kw = dict(rule=0, trace=4000, max_pivots=1500)
    a = oracle.emul_solve(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, climb_budget=-1, **kw)
    b = oracle.emul_solve(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, climb_budget=0, **kw)
    assert a["pivots"] == b["pivots"] == 1500 and b["scans"] > 0
    assert np.array_equal(a["trace"], b["trace"]) and np.array_equal(a["flow"], b["flow"])
    for key in ("parent", "size", "pos", "order", "depth", "psize"):
        assert np.array_equal(a[key], b[key]), key
    assert b["psize"][0] == inst.n + 1 > 65535
    check_tree_invariants(inst.n, b["parent"], b["size"], b["pos"], b["order"], b["depth"], b["psize"])


@pytest.mark.parametrize("price_blocks", [8, 24, 64, 2048])
@pytest.mark.parametrize("shard,shards", [(0, 1), (0, 3), (2, 3), (5, 8)])
def test_incremental_pricing_block_map_matches_the_sweep(price_blocks, shard, shards):
    """The passes that change an arc flag the pricing workgroup that sweeps it (mcf_price_block_of); the map must be
    the sweep's own lane -> arc assignment, for any grid size and any shard, and -1 for other ranks' arcs."""
    from network_flow_solver_amd import generators
    for inst in (generators.netgen_style(300, 2400, seed=5), generators.netgen_style(5000, 70001, seed=6),
                 generators.goto_style(40, 40, seed=7)):
        assert oracle.emul_check_block_map(inst, price_blocks, shard, shards) == 0


def test_devex_pivot_counts_stay_close_to_the_reference():
    """The engine's Devex rule (cyclic block search + the reference's weight reset every 64 basis swaps + its
    block-size tuner) against the reference's own Devex pivot counts (goldens made by running the reference):
    at most 1.2x on every instance.  Without the reset the same rule took 2.4x (netgen_8_12a, round 1)."""
    seen = 0
    for entry, inst in load_synthetic():
        exp = entry["expected"].get("devex")
        if not exp or "iterations" not in exp:
            continue
        res = oracle.emul_solve(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=1)
        assert res["status"] == "optimal" and res["objective"] == int(round(exp["objective"]))
        assert res["pivots"] <= 1.2 * exp["iterations"], (entry["name"], res["pivots"], exp["iterations"])
        seen += 1
    assert seen >= 5


def test_restated_block_selection_tie_rules():
    """oracle.price_block (the restated _select_entering_arc_vectorized, simplex.py:528-617) on a hand-made state:
    first maximum per direction, forward only when strictly greater, weights divide the squared reduced cost."""
    tail = np.array([0, 0, 1, 1, 2, 2], np.int32)
    head = np.array([1, 2, 2, 3, 3, 0], np.int32)
    pot = np.zeros(4)
    inf = np.inf
    #        forward rc=-4   fwd rc=-4    backward rc=+4  bwd rc=+6/w=4   basic       fwd rc=-2
    cost = np.array([-4.0, -4.0, 4.0, 6.0, -9.0, -2.0])
    fwd = np.array([5.0, 5.0, 0.0, 0.0, 5.0, 5.0])
    bwd = np.array([0.0, 0.0, 3.0, 3.0, 0.0, 0.0])
    in_tree = np.array([0, 0, 0, 0, 1, 0], np.uint8)
    w = np.array([1.0, 1.0, 1.0, 4.0, 1.0, 1.0])
    # merits: 16, 16, 16 (backward), 9 (backward), -, 4: forward/backward tie -> backward arc 2
    assert oracle.price_block(tail, head, cost, pot, fwd, bwd, in_tree, w, 0, 6) == (2, -1, 16.0)
    # without the backward candidates: lowest index among the equal forward merits
    assert oracle.price_block(tail, head, cost, pot, fwd, bwd, in_tree, w, 0, 2) == (0, 1, 16.0)
    assert oracle.price_block(tail, head, cost, pot, fwd, bwd, in_tree, w, 1, 2) == (1, 1, 16.0)
    assert oracle.price_block(tail, head, cost, pot, fwd, bwd, in_tree, w, 3, 6) == (3, -1, 9.0)
    assert oracle.price_block(tail, head, cost, pot, fwd, bwd, in_tree, w, 4, 5) is None
    del inf


@pytest.mark.parametrize("rule", [0, 2], ids=["dantzig", "candidate_list"])
def test_key_variants_reach_the_same_optimum(rule):
    """The specialised entering rules as key variants (mcf_core.h: mcf_dantzig_key; specialized_pivots.py:191-424): a different
    pivot order, the same optimum, on every synthetic golden the emulation solves."""
    for entry, inst in load_synthetic()[:8]:
        exp = next(iter(entry["expected"].values()))
        rng = np.random.default_rng(inst.m)
        prio = rng.integers(0, 4, size=inst.m).astype(np.int8)
        counts = set()
        for key_mode, pr in ((0, None), (1, None), (2, prio), (3, None)):
            em = oracle.emul_solve(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=rule | (key_mode << 8), arc_priority=pr)
            assert em["status"] == "optimal" and em["objective"] == int(round(exp["objective"])), (entry["name"], key_mode)
            check_optimality(inst, em["flow"], em["potential"])
            counts.add(em["pivots"])
        assert len(counts) > 1


def test_entering_rule_options_follow_the_reference_dispatch():
    """specialized_pivots.py:452-527: which key variant each structured class gets, and what the priority bytes hold."""
    from network_flow_solver_amd import engine as e
    from network_flow_solver_amd.specializations import NetworkStructure, NetworkType, entering_rule_options

    # a path 0 -> 1 -> 2 plus a node 3 the source does not reach (arc 3 -> 2)
    tail, head = np.array([0, 1, 3], np.int32), np.array([1, 2, 2], np.int32)
    supply = np.array([1, 0, -1, 0], np.int64)
    ids = ["a", "b", "c", "d"]
    st = NetworkStructure(NetworkType.SHORTEST_PATH, False)
    opt = entering_rule_options(st, ids, tail, head, supply)
    assert opt["rule"] == e.RULE_DANTZIG and opt["key_mode"] == e.KEY_PRIORITY
    assert opt["arc_priority"].tolist() == [3, 3, 2]           # backward always; forward only where the tail is labelled
    assert entering_rule_options(st, ids, tail, head, np.array([2, 0, -2, 0], np.int64)) is None     # no unit source: general rule
    assert entering_rule_options(st, ids, tail, head, np.array([2, 0, -2, 0], np.int64), unit=2)["key_mode"] == e.KEY_PRIORITY
    st = NetworkStructure(NetworkType.MAX_FLOW, False)
    assert entering_rule_options(st, ids, tail, head, supply) == {"rule": e.RULE_DANTZIG, "key_mode": e.KEY_CAPACITY}
    assert entering_rule_options(st, ids, tail, head, np.zeros(4, np.int64)) is None
    st = NetworkStructure(NetworkType.BIPARTITE_MATCHING, True, partitions=({"a", "d"}, {"b", "c"}))
    opt = entering_rule_options(st, ids, np.array([0, 3, 0], np.int32), np.array([1, 2, 2], np.int32), np.array([1, -1, -1, 1], np.int64))
    assert opt["key_mode"] == e.KEY_PRIORITY and opt["arc_priority"].tolist() == [1, 1, 1]
    opt = entering_rule_options(st, ids, np.array([0, 3, 0], np.int32), np.array([1, 2, 2], np.int32), np.array([1, -1, -2, 2], np.int64))
    assert opt["arc_priority"].tolist() == [1, 0, 1]            # only unit-supply nodes of the left side count as unmatched
    assert entering_rule_options(NetworkStructure(NetworkType.ASSIGNMENT, True), ids, tail, head, supply)["key_mode"] == e.KEY_FORWARD_FIRST
    assert entering_rule_options(NetworkStructure(NetworkType.TRANSPORTATION, True), ids, tail, head, supply) == {"rule": e.RULE_DANTZIG}
    assert entering_rule_options(NetworkStructure(NetworkType.GENERAL, False), ids, tail, head, supply) is None


# ------------------------------------------------------------------ blocked preorder list (mcf_core.h) == dense preorder array
def _layout(shift: int, pool: int) -> int:
    """rule bits of oracle.emul_solve: 16-19 log2 of the block size, 20-31 the spare blocks (1 = none, k = k - 1, 0 = auto)."""
    return (shift << 16) | (pool << 20)


@pytest.mark.parametrize("rule", [0, 1, 2], ids=["dantzig", "devex_block", "candidate_list"])
@pytest.mark.parametrize("shift,pool", [(2, 0), (2, 1), (3, 5), (4, 0), (6, 0), (6, 1)], ids=lambda x: str(x))
def test_blocked_preorder_list_equals_dense_array(rule, shift, pool):
    """The tree's logical preorder kept in physical blocks with a logical base each (O(subtree + block) element moves per
    basis swap instead of a shift of everything between the subtree's old and new place; replaces the per-pivot BFS
    rebuild basis.py:82-125) is the SAME logical preorder: pivots, flows, potentials, order, positions, sizes and depths
    equal the dense array's for every block size, with a generous pool, a tiny one (frequent dense rewrites into the other
    arena) and none at all (a rewrite on every pivot), by climb and by scan."""
    from network_flow_solver_amd import generators

    cases = [load_synthetic()[0][1], load_synthetic()[4][1], generators.goto_style(12, 12, seed=4), generators.gridgen_style(16, 16, seed=3)]
    for inst in cases:
        for cb in (0, -1):
            ref = oracle.emul_solve(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=rule, climb_budget=cb)
            got = oracle.emul_solve(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=rule | _layout(shift, pool), climb_budget=cb)
            assert got["status"] == ref["status"] == "optimal" and got["objective"] == ref["objective"] and got["pivots"] == ref["pivots"]
            for key in ("flow", "potential", "parent", "pred_arc", "size", "pos", "order", "depth", "psize"):
                assert np.array_equal(got[key], ref[key]), key
            check_tree_invariants(inst.n, got["parent"], got["size"], got["pos"], got["order"], got["depth"], got["psize"])
            if pool == 1:
                assert got["scan_rounds"] > 0.5 * got["pivots"]   # (blocked list: the dense rewrites are reported here)
            # element moves: the re-hung subtrees, at most two cut-off runs of a block per pivot and the dense rewrites --
            # against every position between the old and the new place for the dense array
            assert got["nodes_moved"] <= got["subtree_nodes"] + 2 * (1 << shift) * got["pivots"] + (inst.n + 1) * got["scan_rounds"]


def test_blocked_list_after_every_pivot_and_on_a_warm_start():
    _, inst = load_synthetic()[0]
    for cap in range(1, 60):
        got = oracle.emul_solve(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=2 | _layout(2, 3), max_pivots=cap, climb_budget=0)
        ref = oracle.emul_solve(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=2, max_pivots=cap, climb_budget=0)
        assert np.array_equal(got["order"], ref["order"]) and np.array_equal(got["psize"], ref["psize"]) and np.array_equal(got["pos"], ref["pos"])
        check_tree_invariants(inst.n, got["parent"], got["size"], got["pos"], got["order"], got["depth"], got["psize"])
    cold = oracle.emul_solve(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=0)
    warm = oracle.emul_solve(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=0 | _layout(3, 0), warm_in_tree=cold["in_tree"],
                             warm_at_upper=(cold["flow"] == inst.cap) & (inst.cap > 0) & ~cold["in_tree"].astype(bool))
    assert warm["warm_applied"] and warm["status"] == "optimal" and warm["objective"] == cold["objective"]
    check_tree_invariants(inst.n, warm["parent"], warm["size"], warm["pos"], warm["order"], warm["depth"], warm["psize"])


def test_random_instances_blocked_list_equals_dense_array_and_optimal_bases_need_no_pivots():
    """A small slice of scripts/fuzz_cpu_layout.py and scripts/fuzz_cpu_warm.py (which ran 3 226 / 4 333 cases for
    profiles/r03_fuzz_and_layout_checks.txt): on random seeded instances the emulation on the blocked preorder list (random
    block size, random pool incl. none) equals the emulation on the dense array in everything, and a warm start from the
    optimal basis -- real arcs only, as a Basis hands it over (simplex.py:1744-1765) -- is confirmed in ZERO pivots when the
    real basic arcs span the nodes."""
    import random

    from network_flow_solver_amd import generators

    zero = 0
    for seed in range(5000, 5060):
        rng = random.Random(seed)
        n = rng.choice([12, 40, 90, 250])
        fam = rng.choice(["netgen", "gridgen", "goto"])
        if fam == "netgen":
            inst = generators.netgen_style(n, n * rng.choice([3, 6, 10]), seed=seed)
        else:
            w = max(3, int(n ** 0.5))
            inst = (generators.gridgen_style if fam == "gridgen" else generators.goto_style)(w, w, seed=seed)
        rule = rng.choice([0, 1, 2])
        dense = oracle.emul_solve(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=rule)
        bits = rule | (rng.choice([2, 3, 4, 5]) << 16) | (rng.choice([0, 1, 2, 9]) << 20)   # emul_engine.cpp: decode_rule
        blk = oracle.emul_solve(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=bits)
        assert dense["status"] == blk["status"] and dense["pivots"] == blk["pivots"] and dense["objective"] == blk["objective"], (seed, bits)
        for k in ("flow", "potential", "order", "pos", "psize", "depth", "parent"):
            assert np.array_equal(dense[k], blk[k]), (seed, bits, k)
        if dense["status"] != "optimal":
            continue
        it, au = _basis_of(inst, dense)
        warm = oracle.emul_solve(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=rule, warm_in_tree=it, warm_at_upper=au)
        assert warm["warm_applied"] and warm["status"] == "optimal" and warm["objective"] == dense["objective"], seed
        if inst.n - int(it.sum()) == 1:
            zero += 1
            assert warm["pivots"] == 0, (seed, warm["pivots"])
    assert zero >= 20
