"""CPU-side checks: the C-ABI library loads and exports every symbol include/mcf.h declares,
the product path refuses to run without a GPU (no silent fallback), and the host-side shim
(flattening, option validation, loaders) behaves like the reference's."""

import ctypes
import json
import math

import numpy as np
import pytest

import __graft_entry__ as ge
import network_flow_solver_amd as nfs
from conftest import CASES, ROOT
from network_flow_solver_amd import engine
from network_flow_solver_amd.simplex import flatten_problem


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(str(ge.LIB))
    declared = ge.declared_symbols()
    assert len(declared) >= 15
    assert set(declared) == set(engine.ABI_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert engine.load_library().mcf_abi_version() == engine.ABI_VERSION == 3


def test_struct_layouts_match_the_header():
    text = (ROOT / "include" / "mcf.h").read_text()

    def fields(struct):
        body = text.split(f"typedef struct {struct} {{")[1].split("}")[0]
        out = []
        for line in body.splitlines():
            line = line.split("/*")[0].strip()
            if line.endswith(";") and "(" not in line:
                out.append(line[:-1].split()[-1].split("[")[0])
        return out

    assert fields("mcf_options") == [f for f, _ in engine.McfOptions._fields_]
    assert fields("mcf_stats") == [f for f, _ in engine.McfStats._fields_]


def test_product_path_never_imports_the_oracle():
    for path in (ROOT / "network_flow_solver_amd").rglob("*.py"):
        text = path.read_text()
        assert "import oracle" not in text and "from oracle" not in text, path
    for path in (ROOT / "network_flow_solver_amd" / "csrc").glob("*"):
        if path.suffix in (".hip", ".h", ".cpp"):
            assert "oracle/" not in path.read_text().replace("oracle/emul_engine.cpp", ""), path


@pytest.mark.skipif(engine.load_library().mcf_device_count() > 0, reason="only meaningful without a GPU")
def test_no_gpu_means_loud_failure_not_fallback():
    c = CASES[0]
    problem = nfs.build_problem(c["nodes"], c["arcs"], c["directed"], c["tolerance"])
    with pytest.raises(engine.EngineUnavailableError):
        nfs.solve_min_cost_flow(problem)
    with pytest.raises(engine.EngineUnavailableError):
        engine.McfEngine(2, [0], [1], [1], [1], [1, -1])


def test_bench_without_a_gpu_fails_loudly_and_its_ranks_do_not_wait_for_each_other():
    """`bench.py --gpus 2` without a launcher spawns its two ranks; a rank that dies would leave the other waiting in its
    first collective for ever, so the spawner polls its children and ends the rest as soon as one has failed.  Here (no GPU)
    every rank fails at once: the call must come back promptly with a non-zero exit code and no JSON line -- never a number
    from some CPU path."""
    import subprocess
    import sys
    import time

    if engine.device_count() > 0:
        pytest.skip("a GPU is visible: the failure path is not what runs here")
    for extra in (["--gpus", "2"], []):
        t0 = time.time()
        proc = subprocess.run([sys.executable, str(ROOT / "bench.py"), *extra, "--steps", "5", "--warmup", "1", "--no-cpu-baseline", "--no-hbm-point"],
                              capture_output=True, text=True, timeout=240)
        assert proc.returncode != 0, (extra, proc.stdout[-300:])
        assert '"metric"' not in proc.stdout, proc.stdout[-300:]
        assert time.time() - t0 < 200


def test_flatten_matches_reference_ordering_and_shift():
    case = next(c for c in CASES if c["name"] == "lower_bounds_and_parallel")
    p = nfs.build_problem(case["nodes"], case["arcs"], True, 1e-6)
    f = flatten_problem(p)
    assert f.node_ids == ["a", "b", "c"]
    assert f.keys == [("a", "b"), ("a", "b"), ("a", "c"), ("b", "c")]      # stable sort by (tail, head)
    assert f.cap.tolist() == [7, 4, 5, -1]                                   # capacity - lower; None -> -1
    assert f.lower.tolist() == [3.0, 0.0, 1.0, 2.0]
    assert f.supply.tolist() == [12 - 3 - 1, 3 - 2, -12 + 1 + 2]             # simplex.py:413-415
    assert int(f.supply.sum()) == 0


def test_flatten_scales_decimals_exactly():
    case = next(c for c in CASES if c["name"] == "fractional_costs")
    f = flatten_problem(nfs.build_problem(case["nodes"], case["arcs"], True, 1e-6))
    assert f.flow_scale == 10 and f.cost_scale == 100
    assert f.cost.tolist() == [150, 200, 125] and f.supply.tolist() == [0, 25, -25] or f.supply.sum() == 0
    bad = nfs.build_problem([{"id": "a", "supply": 1.0}, {"id": "b", "supply": -1.0}],
                            [{"tail": "a", "head": "b", "capacity": 2.0, "cost": math.pi}], True, 1e-6)
    with pytest.raises(nfs.SolverConfigurationError):
        flatten_problem(bad)


def test_undirected_expansion_and_validation_errors():
    und = nfs.build_problem([{"id": "u", "supply": 2.0}, {"id": "v", "supply": -2.0}],
                            [{"tail": "u", "head": "v", "capacity": 5.0, "cost": 1.0}], False, 1e-6)
    f = flatten_problem(und)
    assert f.lower.tolist() == [-5.0] and f.cap.tolist() == [10]
    with pytest.raises(nfs.InvalidProblemError):   # infinite-capacity undirected edge, data.py:187-194
        nfs.build_problem([{"id": "u"}, {"id": "v"}], [{"tail": "u", "head": "v", "capacity": None, "cost": 1.0}],
                          False, 1e-6).undirected_expansion()
    with pytest.raises(nfs.InvalidProblemError):   # unbalanced, data.py:141-148
        nfs.build_problem([{"id": "u", "supply": 1.0}, {"id": "v", "supply": 0.0}], [], True, 1e-6)
    with pytest.raises(nfs.InvalidProblemError):   # self loop, data.py:78-82
        nfs.Arc("x", "x", 1.0, 1.0)
    with pytest.raises(nfs.InvalidProblemError):   # duplicate id, data.py:543-546
        nfs.build_problem([{"id": "u"}, {"id": "u"}], [], True, 1e-6)
    with pytest.raises(nfs.InvalidProblemError):   # unknown endpoint, data.py:149-160
        nfs.build_problem([{"id": "u"}], [{"tail": "u", "head": "w", "capacity": 1.0, "cost": 0.0}], True, 1e-6)


def test_solver_options_validation_mirrors_reference():
    o = nfs.SolverOptions()
    assert o.pricing_strategy == "adaptive" and o.tolerance == 1e-6 and o.use_dense_inverse is False
    for kw in ({"tolerance": 0}, {"pricing_strategy": "steepest"}, {"block_size": 0}, {"block_size": "big"},
               {"ft_update_limit": 0}, {"condition_number_threshold": 1}, {"adaptive_ft_min": 300}):
        with pytest.raises(nfs.InvalidProblemError):
            nfs.SolverOptions(**kw)
    assert nfs.SolverOptions(block_size="auto").block_size == "auto"


def test_json_round_trip(tmp_path):
    payload = {"directed": True, "tolerance": 1e-6,
               "nodes": [{"id": "s", "supply": 4.0}, {"id": "t", "supply": -4.0}],
               "arcs": [{"tail": "s", "head": "t", "capacity": 4.0, "cost": 3.0}]}
    path = tmp_path / "p.json"
    path.write_text(json.dumps(payload))
    p = nfs.load_problem(path)
    assert len(p.nodes) == 2 and p.arcs[0].capacity == 4.0 and p.tolerance == 1e-6
    res = nfs.FlowResult(objective=12.0, flows={("s", "t"): 4.0}, status="optimal", iterations=1, duals={"s": 0.0})
    nfs.save_result(tmp_path / "r.json", res)
    saved = json.loads((tmp_path / "r.json").read_text())
    assert saved["flows"] == [{"tail": "s", "head": "t", "flow": 4.0}] and saved["objective"] == 12.0
    (tmp_path / "bad.json").write_text(json.dumps({"nodes": 3}))
    with pytest.raises(nfs.InvalidProblemError):
        nfs.load_problem(tmp_path / "bad.json")


DIMACS = """c demo
p min 3 2
n 1 10
n 3 -10
a 1 2 0 20 1
a 2 3 0 -1 1
"""


def test_dimacs_parser_behaviour(tmp_path):
    p = nfs.parse_dimacs_string(DIMACS)
    assert list(p.nodes) == ["1", "2", "3"] and p.nodes["2"].supply == 0.0 and p.tolerance == 1e-6
    assert p.arcs[1].capacity is None and p.arcs[0].capacity == 20.0
    four = nfs.parse_dimacs_string("p min 2 1\nn 1 1\nn 2 -1\na 1 2 5 7\n")
    assert four.arcs[0].lower == 0.0 and four.arcs[0].capacity == 5.0 and four.arcs[0].cost == 7.0
    for bad in ("n 1 1\n", "p max 2 1\n", "p min 2 2\na 1 2 0 1 1\n", "p min 2 1\na 1 9 0 1 1\n",
                "p min 2 1\nx 1\n", "p min 2 0\np min 2 0\n", "p min 0 0\n"):
        with pytest.raises(nfs.InvalidProblemError):
            nfs.parse_dimacs_string(bad)
    with pytest.raises(FileNotFoundError):
        nfs.parse_dimacs_file(tmp_path / "missing.min")
    f = tmp_path / "x.min"
    f.write_text(DIMACS)
    soa = nfs.parse_dimacs_soa(f)
    assert soa.n == 3 and soa.tail.tolist() == [0, 1] and soa.cap.tolist() == [20, -1] and soa.supply.tolist() == [10, 0, -10]


def test_reference_dimacs_fixtures_parse():
    for c in CASES:
        if "dimacs_text" in c:
            p = nfs.parse_dimacs_string(c["dimacs_text"])
            assert len(p.nodes) == len(c["nodes"]) and len(p.arcs) == len(c["arcs"])


def test_generators_are_deterministic_and_balanced():
    from network_flow_solver_amd import generators as g

    a, b = g.netgen_style(128, 1024, seed=3), g.netgen_style(128, 1024, seed=3)
    assert a.sha256() == b.sha256() and a.m == 1024 and int(a.supply.sum()) == 0
    assert np.all(np.diff(a.tail.astype(np.int64) * a.n + a.head) > 0)        # tail-major, no duplicates
    for inst in (g.gridgen_style(8, 8, 1), g.goto_style(8, 8, 1)):
        assert inst.m == 8 * inst.n and int(inst.supply.sum()) == 0 and (inst.tail != inst.head).all()


def test_solver_adapter_never_raises_and_reports_availability():
    from network_flow_solver_amd.adapter import Mi355xAdapter, SolverResult

    assert Mi355xAdapter.name == "network_solver_mi355x" and isinstance(Mi355xAdapter.get_version(), str)
    assert Mi355xAdapter.is_available() == (engine.load_library().mcf_device_count() > 0)
    c = CASES[0]
    res = Mi355xAdapter.solve(nfs.build_problem(c["nodes"], c["arcs"], c["directed"], c["tolerance"]))
    assert isinstance(res, SolverResult)
    if not Mi355xAdapter.is_available():
        assert res.status == "error" and "EngineUnavailableError" in res.error_message   # loud, but not an exception


def test_native_dimacs_reader_matches_object_parser(tmp_path):
    """mcf_dimacs_scan / mcf_dimacs_load (C, host side of the library) against the reference-shaped
    object parser on generator output and on the reference's three .min fixtures, plus the error cases."""
    from network_flow_solver_amd import generators as g

    inst = g.netgen_style(200, 1600, seed=5)
    inst.cap[::7] = -1                                         # some uncapacitated arcs
    path = tmp_path / "net.min"
    g.write_dimacs(inst, path)
    soa = nfs.parse_dimacs_soa(path)
    assert soa.n == inst.n and np.array_equal(soa.tail, inst.tail) and np.array_equal(soa.head, inst.head)
    assert np.array_equal(soa.cost, inst.cost) and np.array_equal(soa.cap, inst.cap) and np.array_equal(soa.supply, inst.supply)
    obj = nfs.parse_dimacs_file(path)
    assert len(obj.arcs) == soa.m and [a.capacity for a in obj.arcs[:8]] == [None if c < 0 else float(c) for c in soa.cap[:8]]
    for c in CASES:
        if "dimacs_text" in c:
            f = tmp_path / (c["name"] + ".min")
            f.write_text(c["dimacs_text"])
            s2 = nfs.parse_dimacs_soa(f)
            p2 = nfs.parse_dimacs_file(f)
            assert s2.n == len(p2.nodes) and s2.m == len(p2.arcs)
            assert [int(x) for x in s2.cost] == [int(a.cost) for a in p2.arcs]
    variants = {"four.min": "p min 2 1\nn 1 1\nn 2 -1\na 1 2 5 7\n", "inf.min": "c x\np min 2 2\nn 1 1\nn 2 -1\na 1 2 0 inf 3\na 1 2 0 1e15 4\n"}
    for name, text in variants.items():
        (tmp_path / name).write_text(text)
    four = nfs.parse_dimacs_soa(tmp_path / "four.min")
    assert four.cap.tolist() == [5] and four.cost.tolist() == [7]
    assert nfs.parse_dimacs_soa(tmp_path / "inf.min").cap.tolist() == [-1, -1]
    bad = {"nop.min": "n 1 1\n", "max.min": "p max 2 1\n", "count.min": "p min 2 2\na 1 2 0 1 1\n",
           "range.min": "p min 2 1\na 1 9 0 1 1\n", "kind.min": "p min 2 1\nx 1\n", "frac.min": "p min 2 1\na 1 2 0 1 1.5\n",
           "short.min": "p min 2 1\nn 1 1\nn 2 -1\na 1 2 6 5 7\n", "loop.min": "p min 2 1\na 1 1 0 5 7\n"}
    for name, text in bad.items():
        (tmp_path / name).write_text(text)
        with pytest.raises(nfs.InvalidProblemError):
            nfs.parse_dimacs_soa(tmp_path / name)
    # lower bounds travel through the native reader; the shim applies the reference's shift (simplex.py:413-428)
    (tmp_path / "lower.min").write_text("p min 3 2\nn 1 4\nn 3 -4\na 1 2 1 5 7\na 2 3 2 6 1\n")
    low = nfs.parse_dimacs_soa(tmp_path / "lower.min")
    assert low.lower.tolist() == [1, 2] and low.capacity.tolist() == [5, 6]
    from network_flow_solver_amd.simplex import flatten_soa
    f = flatten_soa(low)
    assert f.cap.tolist() == [4, 4] and f.supply.tolist() == [3, -1, -2] and f.soa and f.keys[1] == ("2", "3")
    # the same object behind the NetworkProblem surface
    assert isinstance(low, nfs.SoAProblem) and low.directed and len(low.nodes) == 3 and low.arcs[1].lower == 2.0
    obj = nfs.parse_dimacs_file(tmp_path / "lower.min")                     # small file: the object model, as in the reference
    assert isinstance(obj, nfs.NetworkProblem) and [a.lower for a in obj.arcs] == [1.0, 2.0]
    assert isinstance(nfs.parse_dimacs_file(tmp_path / "lower.min", native=True), nfs.SoAProblem)


def test_lazy_result_views_behave_like_the_reference_dicts():
    from network_flow_solver_amd.data import ArrayBasis, LazyDuals, LazyFlows

    tail = np.array([0, 0, 1, 0], np.int32)
    head = np.array([1, 1, 2, 2], np.int32)
    flows = LazyFlows(tail, head, np.array([3, 2, 5, 0], np.int64), 1e-6)
    assert flows == {("1", "2"): 5.0, ("2", "3"): 5.0} and len(flows) == 2 and ("1", "3") not in flows   # parallel arcs summed
    assert dict(flows.items())[("2", "3")] == 5.0 and sorted(flows) == [("1", "2"), ("2", "3")]
    duals = LazyDuals(np.array([0.0, -3.0, 7.5]))
    assert duals["2"] == -3.0 and len(duals) == 3 and list(duals)[:2] == ["1", "2"]
    b = ArrayBasis(tail, head, np.array([1, 0, 1, 0]), np.array([0, 1, 0, 0]), np.array([3, 2, 5, 0], np.int64))
    assert b.tree_arcs == {("1", "2"), ("2", "3")} and b.arc_flows[("2", "3")] == 5.0


def test_structure_analysis_matches_the_reference_classification():
    """specializations.analyze_network_structure (array based) against the class the reference itself assigned to
    every fixture (``network_type`` in tests/golden/cases.json, recorded by make_golden.py), for the object model and
    for the flat SoAProblem form of the DIMACS fixtures."""
    from network_flow_solver_amd.specializations import NetworkType, analyze_network_structure

    seen = set()
    for c in CASES:
        p = nfs.build_problem(c["nodes"], c["arcs"], c["directed"], c["tolerance"])
        st = analyze_network_structure(p)
        assert st.network_type.value == c["network_type"], c["name"]
        seen.add(st.network_type)
        if st.network_type in (NetworkType.TRANSPORTATION, NetworkType.ASSIGNMENT):
            assert st.is_bipartite and st.partitions is not None and not st.transshipment_nodes
    assert {NetworkType.GENERAL, NetworkType.TRANSPORTATION, NetworkType.ASSIGNMENT, NetworkType.SHORTEST_PATH,
            NetworkType.MAX_FLOW, NetworkType.BIPARTITE_MATCHING} <= seen
    soa = nfs.SoAProblem(4, [0, 0, 1, 1], [2, 3, 2, 3], [4, 6, 3, 5], [10, 10, 12, 15], [10, 15, -12, -13])
    assert analyze_network_structure(soa).network_type is NetworkType.TRANSPORTATION
    soa = nfs.SoAProblem(4, [0, 0, 1, 1], [2, 3, 2, 3], [4, 6, 3, 5], [1, 1, 1, 1], [1, 1, -1, -1])
    assert analyze_network_structure(soa).network_type is NetworkType.ASSIGNMENT
