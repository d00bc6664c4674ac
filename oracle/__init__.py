"""oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes front end of ``oracle/ref_simplex.c``, the CPU restatement of the
reference's network-simplex path.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg import this package; nothing under
``network_flow_solver_amd/`` does.

Parity pin: every fixture in ``tests/golden/`` was produced by running the
reference itself (``tests/golden/make_golden.py``) and
``tests/test_oracle_golden.py`` checks this oracle against all of them
(status, objective, flows; Dantzig iteration counts too).
"""

from __future__ import annotations

import ctypes
import math
import os
import subprocess
from dataclasses import dataclass, field
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "libref_simplex.so"
_SRC_PATH = _HERE / "ref_simplex.c"

STRATEGIES = {"dantzig": 0, "devex": 1, "candidate_list": 2, "adaptive": 3}
STATUS_NAMES = {0: "optimal", 1: "infeasible", 2: "iteration_limit", 3: "unbounded"}


def build(force: bool = False) -> Path:
    """Compile the C restatement with gcc (seconds)."""
    if force or not _LIB_PATH.exists() or _LIB_PATH.stat().st_mtime < _SRC_PATH.stat().st_mtime:
        cmd = ["gcc", "-O2", "-std=gnu11", "-D_GNU_SOURCE", "-fPIC", "-shared",
               "-o", str(_LIB_PATH), str(_SRC_PATH), "-lm"]
        subprocess.run(cmd, check=True, cwd=str(_HERE))
    return _LIB_PATH


_lib = None


def _load():
    global _lib
    if _lib is None:
        build()
        lib = ctypes.CDLL(str(_LIB_PATH))
        i32p = ctypes.POINTER(ctypes.c_int32)
        f64p = ctypes.POINTER(ctypes.c_double)
        u8p = ctypes.POINTER(ctypes.c_uint8)
        i64p = ctypes.POINTER(ctypes.c_int64)
        lib.ref_solve.restype = ctypes.c_int
        lib.ref_solve.argtypes = [
            ctypes.c_int, ctypes.c_int64, i32p, i32p, f64p, f64p, f64p, f64p, ctypes.c_double,
            ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, u8p,
            ctypes.POINTER(ctypes.c_int), f64p, f64p, f64p, u8p, i64p,
        ]
        lib.ref_price_dantzig.restype = ctypes.c_int64
        lib.ref_price_dantzig.argtypes = [
            ctypes.c_int64, i32p, i32p, f64p, f64p, f64p, f64p, u8p, ctypes.c_double, ctypes.c_int,
            ctypes.POINTER(ctypes.c_int),
        ]
        lib.ref_price_block.restype = ctypes.c_int64
        lib.ref_price_block.argtypes = [
            ctypes.c_int64, i32p, i32p, f64p, f64p, f64p, f64p, u8p, f64p, ctypes.c_int64, ctypes.c_int64,
            ctypes.c_double, ctypes.c_int, ctypes.c_int64, ctypes.POINTER(ctypes.c_int), f64p,
        ]
        _lib = lib
    return _lib


def _ptr(a: np.ndarray, ctype):
    return a.ctypes.data_as(ctypes.POINTER(ctype))


@dataclass
class OracleResult:
    status: str
    objective: float
    flows: dict = field(default_factory=dict)  # (tail_id, head_id) -> flow, reference post-processing applied
    iterations: int = 0
    duals: dict = field(default_factory=dict)
    degenerate_pivots: int = 0
    arcs_priced: int = 0
    arc_flow: np.ndarray | None = None  # per input arc (reference internal order), flow + shift
    unbounded_arc: tuple | None = None
    solve_seconds: float = 0.0
    # smallest |reduced cost| over non-basic arcs: > 0 means the optimal flow is unique
    # (dual non-degenerate); ~0 means alternative optima may exist
    min_nonbasic_abs_rc: float = math.inf


def solve_arrays(
    n: int,
    tail1: np.ndarray,
    head1: np.ndarray,
    cost: np.ndarray,
    cap: np.ndarray,
    lower: np.ndarray | None,
    supply: np.ndarray,
    tolerance: float = 1e-6,
    strategy: str = "dantzig",
    use_vectorized_pricing: bool = True,
    block_size: int | None = None,
    max_iterations: int | None = None,
    pivot_budget: int | None = None,
    special: int = 0,
    left_part: np.ndarray | None = None,
):
    """Raw call: nodes 1..n (0 is the root), arcs in reference-internal order.
    ``special``: SPECIAL_TYPES value of the specialised pivot strategy the reference would install
    (specialized_pivots.py:452-527); ``left_part``: uint8[n+1] left partition for bipartite matching.

    Returns (status, objective, flow[m], potential[n+1], in_tree[m], stats[5], seconds).
    """
    import time

    lib = _load()
    m = int(tail1.shape[0])
    tail1 = np.ascontiguousarray(tail1, dtype=np.int32)
    head1 = np.ascontiguousarray(head1, dtype=np.int32)
    cost = np.ascontiguousarray(cost, dtype=np.float64)
    cap = np.ascontiguousarray(cap, dtype=np.float64)
    lower_a = np.ascontiguousarray(lower if lower is not None else np.zeros(m), dtype=np.float64)
    supply = np.ascontiguousarray(supply, dtype=np.float64)
    assert supply.shape[0] == n
    status = ctypes.c_int(0)
    objective = ctypes.c_double(0.0)
    flow = np.zeros(max(m, 1), dtype=np.float64)
    pot = np.zeros(n + 1, dtype=np.float64)
    in_tree = np.zeros(max(m, 1), dtype=np.uint8)
    stats = np.zeros(8, dtype=np.int64)
    t0 = time.perf_counter()
    rc = lib.ref_solve(
        n, m, _ptr(tail1, ctypes.c_int32), _ptr(head1, ctypes.c_int32), _ptr(cost, ctypes.c_double),
        _ptr(cap, ctypes.c_double), _ptr(lower_a, ctypes.c_double), _ptr(supply, ctypes.c_double),
        float(tolerance), STRATEGIES[strategy], 1 if use_vectorized_pricing else 0,
        int(block_size) if block_size else 0,
        -1 if max_iterations is None else int(max_iterations),
        -1 if pivot_budget is None else int(pivot_budget),
        int(special), None if left_part is None else _ptr(np.ascontiguousarray(left_part, np.uint8), ctypes.c_uint8),
        ctypes.byref(status), ctypes.byref(objective), _ptr(flow, ctypes.c_double),
        _ptr(pot, ctypes.c_double), _ptr(in_tree, ctypes.c_uint8), _ptr(stats, ctypes.c_int64),
    )
    dt = time.perf_counter() - t0
    if rc != 0:
        raise RuntimeError(f"ref_solve failed with code {rc}")
    return status.value, objective.value, flow[:m], pot, in_tree[:m], stats, dt


SPECIAL_TYPES = {"general": 0, "transportation": 1, "assignment": 2, "bipartite_matching": 3, "max_flow": 4,
                 "shortest_path": 5}


def detect_network_type(nodes: list[dict], arcs: list[dict], directed: bool, tolerance: float):
    """specializations.py:60-288 restated on plain dicts: (type name, left partition ids or None).
    Source / sink / transshipment split by the PROBLEM tolerance, BFS 2-colouring over the undirected arc graph
    (components started in node order; colour 0 = left), then the reference's order of type tests."""
    ids = [str(nd["id"]) for nd in nodes]
    supply = {str(nd["id"]): float(nd.get("supply", 0.0)) for nd in nodes}
    sources = {i for i in ids if supply[i] > tolerance}
    sinks = {i for i in ids if supply[i] < -tolerance}
    transship = len(ids) - len(sources) - len(sinks)
    total_supply = sum(supply[i] for i in ids if supply[i] > tolerance)
    total_demand = sum(abs(supply[i]) for i in ids if supply[i] < -tolerance)
    balanced = abs(total_supply - total_demand) <= tolerance
    has_lower = any(float(a.get("lower", 0.0)) > tolerance for a in arcs)
    adj: dict[str, list[str]] = {i: [] for i in ids}
    for a in arcs:
        adj[str(a["tail"])].append(str(a["head"]))
        adj[str(a["head"])].append(str(a["tail"]))
    colour: dict[str, int] = {}
    bipartite = bool(ids)
    for start in ids:
        if not bipartite:
            break
        if start in colour:
            continue
        colour[start] = 0
        queue = [start]
        while queue and bipartite:
            node = queue.pop(0)
            for nb in adj[node]:
                if nb not in colour:
                    colour[nb] = 1 - colour[node]
                    queue.append(nb)
                elif colour[nb] == colour[node]:
                    bipartite = False
                    break
    left = {i for i, c in colour.items() if c == 0} if bipartite else None
    ns, nk = len(sources), len(sinks)
    if transship == 0 and ns > 0 and nk > 0 and bipartite and not has_lower:
        if all(str(a["tail"]) in sources and str(a["head"]) in sinks for a in arcs):
            if balanced and ns == nk and all(abs(supply[i] - 1.0) <= tolerance for i in sources) \
                    and all(abs(supply[i] + 1.0) <= tolerance for i in sinks):
                return "assignment", left
            return "transportation", left
    if ns == 1 and nk == 1:
        so, si = next(iter(sources)), next(iter(sinks))
        if abs(supply[so] - 1.0) <= tolerance and abs(supply[si] + 1.0) <= tolerance:
            return "shortest_path", left
    if bipartite and not has_lower:
        if all(abs(abs(supply[i]) - 1.0) <= tolerance or abs(supply[i]) <= tolerance for i in ids):
            return "bipartite_matching", left
    if ns == 1 and nk == 1 and not has_lower:
        costs = [float(a.get("cost", 0.0)) for a in arcs]
        if all(abs(c) <= tolerance for c in costs) or all(abs(c - 1.0) <= tolerance for c in costs):
            return "max_flow", left
    return "general", left


def solve_dicts(
    nodes: list[dict],
    arcs: list[dict],
    directed: bool = True,
    tolerance: float = 1e-6,
    strategy: str = "dantzig",
    max_iterations: int | None = None,
    **kw,
) -> OracleResult:
    """Solve a reference-style problem description the way the reference would.

    Mirrors ``NetworkSimplex.__init__`` ordering: node ids sorted as strings
    (simplex.py:149), arcs sorted by (tail, head) strings (simplex.py:395),
    undirected edges expanded to lower = -capacity (data.py:162-223); the network
    type decides which specialised pivot strategy runs first (simplex.py:133-137, 259-261).
    """
    ids = sorted(str(nd["id"]) for nd in nodes)
    index = {nid: i + 1 for i, nid in enumerate(ids)}
    if "special" not in kw:
        kind, left = detect_network_type(nodes, arcs, directed, tolerance)
        kw["special"] = SPECIAL_TYPES[kind]
        if left is not None and kind == "bipartite_matching":
            lp = np.zeros(len(ids) + 1, dtype=np.uint8)
            for nid in left:
                lp[index[nid]] = 1
            kw["left_part"] = lp
    supply = np.zeros(len(ids), dtype=np.float64)
    for nd in nodes:
        supply[index[str(nd["id"])] - 1] = float(nd.get("supply", 0.0))
    recs = []
    for a in arcs:
        capv = a.get("capacity")
        lowerv = float(a.get("lower", 0.0))
        if not directed:
            lowerv = -float(capv)
        recs.append((str(a["tail"]), str(a["head"]), float(a.get("cost", 0.0)),
                     math.inf if capv is None else float(capv), lowerv))
    recs.sort(key=lambda r: (r[0], r[1]))  # stable, like list.sort in the reference
    m = len(recs)
    tail1 = np.array([index[r[0]] for r in recs], dtype=np.int32).reshape(m)
    head1 = np.array([index[r[1]] for r in recs], dtype=np.int32).reshape(m)
    cost = np.array([r[2] for r in recs], dtype=np.float64).reshape(m)
    cap = np.array([r[3] for r in recs], dtype=np.float64).reshape(m)
    lower = np.array([r[4] for r in recs], dtype=np.float64).reshape(m)
    st, obj, flow, pot, in_tree, stats, dt = solve_arrays(
        len(ids), tail1, head1, cost, cap, lower, supply, tolerance, strategy,
        max_iterations=max_iterations, **kw)
    status = STATUS_NAMES[st]
    res = OracleResult(status=status, objective=float(round(obj, 12)), iterations=int(stats[0]),
                       degenerate_pivots=int(stats[1]), arcs_priced=int(stats[2]), solve_seconds=dt)
    if status == "unbounded":
        ua = int(stats[3])
        res.unbounded_arc = (recs[ua][0], recs[ua][1]) if 0 <= ua < m else None
        return res
    if stats[4]:  # infeasible / phase-1 iteration limit: FlowResult(objective=0.0, flows={}, duals={})
        res.objective = 0.0
        return res
    flows: dict = {}
    for i, r in enumerate(recs):  # simplex.py:1703-1721
        key = (r[0], r[1])
        flows[key] = flows.get(key, 0.0) + float(flow[i])
    res.flows = {k: float(round(v, 12)) for k, v in flows.items() if abs(v) > tolerance}
    res.duals = {nid: float(round(pot[index[nid]], 12)) for nid in ids}
    res.arc_flow = flow
    if m:
        rc = cost + pot[tail1] - pot[head1]
        nonbasic = in_tree == 0
        if nonbasic.any():
            res.min_nonbasic_abs_rc = float(np.abs(rc[nonbasic]).min())
    return res


def solve_soa(inst, strategy: str = "dantzig", reference_order: bool = True, tolerance: float = 1e-6,
              pivot_budget: int | None = None, **kw):
    """Solve a generators.ArcSoA instance (DIMACS ids "1".."n").

    ``reference_order=True`` applies the reference's string sort of node ids and
    arc keys so pivots follow the reference; False keeps numeric order (cheaper
    to set up for the 1M-node cpu_baseline sample; same optimum).
    Returns dict(status, objective, iterations, arcs_priced, flow (input arc order), seconds).
    """
    n, m = inst.n, inst.m
    if reference_order:
        ids = np.array([str(v + 1) for v in range(n)])
        order = np.argsort(ids, kind="stable")  # lexicographic, like sorted(str)
        rank = np.empty(n, dtype=np.int64)
        rank[order] = np.arange(n)
        t_s, h_s = ids[inst.tail], ids[inst.head]
        arc_order = np.lexsort((h_s, t_s))  # primary tail string, then head string; stable
        node1 = rank + 1
    else:
        arc_order = np.arange(m)
        node1 = np.arange(1, n + 1)
    tail1 = node1[inst.tail[arc_order]].astype(np.int32)
    head1 = node1[inst.head[arc_order]].astype(np.int32)
    cap = inst.cap[arc_order].astype(np.float64)
    cap[cap < 0] = np.inf
    supply = np.zeros(n, dtype=np.float64)
    supply[node1 - 1] = inst.supply.astype(np.float64)
    st, obj, flow, pot, in_tree, stats, dt = solve_arrays(
        n, tail1, head1, inst.cost[arc_order].astype(np.float64), cap, None, supply, tolerance, strategy,
        pivot_budget=pivot_budget, **kw)
    flow_in = np.empty(m, dtype=np.float64)
    flow_in[arc_order] = flow
    return {"status": STATUS_NAMES[st], "objective": float(round(obj, 12)), "iterations": int(stats[0]),
            "degenerate_pivots": int(stats[1]), "arcs_priced": int(stats[2]), "flow": flow_in, "seconds": dt}


def price_dantzig(tail, head, cost, potential, fwd_res, bwd_res, in_tree, tolerance=1e-6, allow_zero=False):
    """One full-scan Dantzig pass (simplex_pricing.py:97-137). Returns (arc, dir) or None."""
    lib = _load()
    m = int(len(tail))
    d = ctypes.c_int(0)
    arc = lib.ref_price_dantzig(
        m, _ptr(np.ascontiguousarray(tail, np.int32), ctypes.c_int32),
        _ptr(np.ascontiguousarray(head, np.int32), ctypes.c_int32),
        _ptr(np.ascontiguousarray(cost, np.float64), ctypes.c_double),
        _ptr(np.ascontiguousarray(potential, np.float64), ctypes.c_double),
        _ptr(np.ascontiguousarray(fwd_res, np.float64), ctypes.c_double),
        _ptr(np.ascontiguousarray(bwd_res, np.float64), ctypes.c_double),
        _ptr(np.ascontiguousarray(in_tree, np.uint8), ctypes.c_uint8),
        float(tolerance), 1 if allow_zero else 0, ctypes.byref(d))
    return None if arc < 0 else (int(arc), int(d.value))


def price_block(tail, head, cost, potential, fwd_res, bwd_res, in_tree, weights, start, end, tolerance=1e-6,
                allow_zero=False, excluded=-1):
    """One vectorised block selection (NetworkSimplex._select_entering_arc_vectorized, simplex.py:528-617) over
    arcs [start, end).  Returns (arc, dir, merit) or None."""
    lib = _load()
    m = int(len(tail))
    d = ctypes.c_int(0)
    merit = ctypes.c_double(0.0)
    arc = lib.ref_price_block(
        m, _ptr(np.ascontiguousarray(tail, np.int32), ctypes.c_int32),
        _ptr(np.ascontiguousarray(head, np.int32), ctypes.c_int32),
        _ptr(np.ascontiguousarray(cost, np.float64), ctypes.c_double),
        _ptr(np.ascontiguousarray(potential, np.float64), ctypes.c_double),
        _ptr(np.ascontiguousarray(fwd_res, np.float64), ctypes.c_double),
        _ptr(np.ascontiguousarray(bwd_res, np.float64), ctypes.c_double),
        _ptr(np.ascontiguousarray(in_tree, np.uint8), ctypes.c_uint8),
        _ptr(np.ascontiguousarray(weights, np.float64), ctypes.c_double),
        int(start), int(end), float(tolerance), 1 if allow_zero else 0, int(excluded), ctypes.byref(d),
        ctypes.byref(merit))
    return None if arc < 0 else (int(arc), int(d.value), float(merit.value))


# ---------------------------------------------------------------------------
# CPU emulation of the HIP engine's own pivot algorithm (oracle/emul_engine.cpp).
# Same headers as the kernels, scalar loops instead of kernels.  Test-only.
# ---------------------------------------------------------------------------
_EMUL_LIB_PATH = _HERE / "libmcf_emul.so"
_EMUL_SRC = _HERE / "emul_engine.cpp"
_CORE_HEADERS = [
    _HERE.parent / "network_flow_solver_amd" / "csrc" / "mcf_core.h",
    _HERE.parent / "network_flow_solver_amd" / "csrc" / "mcf_host.h",
]


def build_emul(force: bool = False) -> Path:
    newest = max(p.stat().st_mtime for p in [_EMUL_SRC, *_CORE_HEADERS])
    if force or not _EMUL_LIB_PATH.exists() or _EMUL_LIB_PATH.stat().st_mtime < newest:
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", str(_EMUL_LIB_PATH), str(_EMUL_SRC)]
        subprocess.run(cmd, check=True, cwd=str(_HERE))
    return _EMUL_LIB_PATH


_emul = None


def _load_emul():
    global _emul
    if _emul is None:
        build_emul()
        lib = ctypes.CDLL(str(_EMUL_LIB_PATH))
        i32p = ctypes.POINTER(ctypes.c_int32)
        i64p = ctypes.POINTER(ctypes.c_int64)
        i8p = ctypes.POINTER(ctypes.c_int8)
        lib.emul_solve.restype = ctypes.c_int
        lib.emul_solve.argtypes = [
            ctypes.c_int32, ctypes.c_int64, i32p, i32p, i64p, i64p, i64p, ctypes.c_int32, ctypes.c_int64,
            ctypes.c_int64, i32p, i64p, i64p, i64p, i8p, i64p, i32p, i32p, i32p, i32p, i32p, i64p, ctypes.c_int64,
            ctypes.c_int32, i32p, i32p, ctypes.c_int32, i64p, i8p, i8p, i32p, i8p,
        ]
        _emul = lib
    return _emul


def emul_check_block_map(inst, price_blocks: int, shard: int = 0, shards: int = 1) -> int:
    """Arcs on which mcf_price_block_of (incremental pricing's arc -> pricing-workgroup map) disagrees with the
    sweep's loop structure; 0 = consistent."""
    lib = _load_emul()
    i32p, i64p = ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int64)
    lib.emul_check_block_map.argtypes = [ctypes.c_int32, ctypes.c_int64, i32p, i32p, i64p, i64p, i64p, ctypes.c_int32,
                                         ctypes.c_int64, ctypes.c_int64]
    lib.emul_check_block_map.restype = ctypes.c_int64
    tail = np.ascontiguousarray(inst.tail, np.int32); head = np.ascontiguousarray(inst.head, np.int32)
    cost = np.ascontiguousarray(inst.cost, np.int64); cap = np.ascontiguousarray(inst.cap, np.int64)
    supply = np.ascontiguousarray(inst.supply, np.int64)
    return int(lib.emul_check_block_map(inst.n, len(tail), _ptr(tail, ctypes.c_int32), _ptr(head, ctypes.c_int32),
                                        _ptr(cost, ctypes.c_int64), _ptr(cap, ctypes.c_int64), _ptr(supply, ctypes.c_int64),
                                        price_blocks, shard, shards))


def emul_solve(n, tail, head, cost, cap, supply, rule: int = 0, block_size: int = 0, max_pivots: int = -1,
               trace: int = 0, bucketed: bool = True, climb_budget: int = -1,
               warm_in_tree=None, warm_at_upper=None, arc_priority=None) -> dict:
    """Run the engine's integer pivot algorithm on the CPU. Arrays are 0-based ints; cap < 0 = inf.
    climb_budget < 0: the cycle is always found by pointer chasing; k >= 0: after k round trips the
    position-space scan (mcf_pivot_scan) takes over.  Bits 8-9 of `rule`: mcf_options.key_mode (1 forward first,
    2 arc_priority, 3 capacity-weighted)."""
    lib = _load_emul()
    m = int(len(tail))
    tail = np.ascontiguousarray(tail, np.int32)
    head = np.ascontiguousarray(head, np.int32)
    cost = np.ascontiguousarray(cost, np.int64)
    cap = np.ascontiguousarray(cap, np.int64)
    supply = np.ascontiguousarray(supply, np.int64)
    status = ctypes.c_int32(0)
    obj = np.zeros(2, np.int64)
    flow = np.zeros(max(m, 1), np.int64)
    pot = np.zeros(n, np.int64)
    in_tree = np.zeros(max(m, 1), np.int8)
    stats = np.zeros(12, np.int64)
    parent, pred, size, pos, order, depth, psize = (np.zeros(n + 1, np.int32) for _ in range(7))
    tr = np.full(max(trace, 1), -2, np.int64)
    scan_stats = np.zeros(2, np.int64)
    wt = None if warm_in_tree is None else np.ascontiguousarray(warm_in_tree, np.int8)
    wu = None if warm_at_upper is None else np.ascontiguousarray(warm_at_upper, np.int8)
    warm_applied = ctypes.c_int32(0)
    pr = None if arc_priority is None else np.ascontiguousarray(arc_priority, np.int8)
    rc = lib.emul_solve(
        n, m, _ptr(tail, ctypes.c_int32), _ptr(head, ctypes.c_int32), _ptr(cost, ctypes.c_int64),
        _ptr(cap, ctypes.c_int64), _ptr(supply, ctypes.c_int64), rule, block_size, max_pivots,
        ctypes.byref(status), _ptr(obj, ctypes.c_int64), _ptr(flow, ctypes.c_int64), _ptr(pot, ctypes.c_int64),
        _ptr(in_tree, ctypes.c_int8), _ptr(stats, ctypes.c_int64), _ptr(parent, ctypes.c_int32),
        _ptr(pred, ctypes.c_int32), _ptr(size, ctypes.c_int32), _ptr(pos, ctypes.c_int32),
        _ptr(order, ctypes.c_int32), _ptr(tr, ctypes.c_int64), trace, 1 if bucketed else 0, _ptr(depth, ctypes.c_int32),
        _ptr(psize, ctypes.c_int32), climb_budget, _ptr(scan_stats, ctypes.c_int64),
        None if wt is None else _ptr(wt, ctypes.c_int8), None if wu is None else _ptr(wu, ctypes.c_int8), ctypes.byref(warm_applied),
        None if pr is None else _ptr(pr, ctypes.c_int8))
    if rc != 0:
        raise RuntimeError(f"emul_solve failed with code {rc}")
    objective = (int(obj[0]) << 64) + (int(obj[1]) & ((1 << 64) - 1))
    return {
        "status": STATUS_NAMES[status.value], "objective": objective, "flow": flow[:m], "potential": pot,
        "in_tree": in_tree[:m], "pivots": int(stats[0]), "degenerate": int(stats[1]), "bound_flips": int(stats[2]),
        "arcs_priced": int(stats[3]), "nodes_moved": int(stats[4]), "subtree_nodes": int(stats[5]),
        "cycle_arcs": int(stats[6]), "unbounded_arc": int(stats[7]), "artificial_flow": int(stats[8]),
        "seconds": stats[9] / 1e9, "minor_pivots": int(stats[10]), "major_sweeps": int(stats[11]), "parent": parent, "pred_arc": pred, "size": size, "pos": pos, "order": order, "depth": depth, "psize": psize, "scans": int(scan_stats[0]), "scan_rounds": int(scan_stats[1]), "warm_applied": bool(warm_applied.value),
        "trace": tr[:trace] if trace else None,
    }


class EmulStepper:
    """Step-wise handle on the CPU emulation (one replica): price a shard, apply a pivot.
    Mirrors how the HIP engine is driven per pivot in the arc-sharded multi-GPU loop."""

    def __init__(self, n, tail, head, cost, cap, supply, rule: int = 0, block_size: int = 0, bucketed: bool = True, arc_priority=None):
        lib = _load_emul()
        i32p, i64p = ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int64)
        lib.emul_create.restype = ctypes.c_void_p
        lib.emul_create.argtypes = [ctypes.c_int32, ctypes.c_int64, i32p, i32p, i64p, i64p, i64p, ctypes.c_int32, ctypes.c_int64,
                                    ctypes.c_int32, ctypes.POINTER(ctypes.c_int8)]
        lib.emul_price.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, i64p]
        lib.emul_price.restype = None
        lib.emul_pivot.argtypes = [ctypes.c_void_p, i64p, ctypes.c_int32]
        lib.emul_pivot.restype = None
        lib.emul_set_max_pivots.argtypes = [ctypes.c_void_p, ctypes.c_int64]
        lib.emul_set_max_pivots.restype = None
        lib.emul_poll.argtypes = [ctypes.c_void_p, i32p, i64p, i64p, i64p]
        lib.emul_poll.restype = None
        lib.emul_destroy.argtypes = [ctypes.c_void_p]
        lib.emul_destroy.restype = None
        lib.emul_set_shards.argtypes = [ctypes.c_void_p, ctypes.c_int64]
        lib.emul_set_shards.restype = None
        lib.emul_list_len.argtypes = [ctypes.c_void_p]
        lib.emul_list_len.restype = ctypes.c_int32
        lib.emul_minor_cap.argtypes = [ctypes.c_void_p]
        lib.emul_minor_cap.restype = ctypes.c_int32
        lib.emul_price_list.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, i64p]
        lib.emul_price_list.restype = None
        lib.emul_pivots.argtypes = [ctypes.c_void_p, i64p, ctypes.c_int32, ctypes.c_int32]
        lib.emul_pivots.restype = None
        self._lib = lib
        self.m = int(len(tail))
        self._keep = [np.ascontiguousarray(tail, np.int32), np.ascontiguousarray(head, np.int32),
                      np.ascontiguousarray(cost, np.int64), np.ascontiguousarray(cap, np.int64),
                      np.ascontiguousarray(supply, np.int64)]
        t, h, c, cp, s = self._keep
        pr = None if arc_priority is None else np.ascontiguousarray(arc_priority, np.int8)
        self._h = lib.emul_create(int(n), self.m, _ptr(t, ctypes.c_int32), _ptr(h, ctypes.c_int32),
                                  _ptr(c, ctypes.c_int64), _ptr(cp, ctypes.c_int64), _ptr(s, ctypes.c_int64),
                                  int(rule), int(block_size), 1 if bucketed else 0,
                                  None if pr is None else _ptr(pr, ctypes.c_int8))
        if not self._h:
            raise RuntimeError("emul_create failed")

    def price(self, shard: int, shards: int, out: np.ndarray) -> None:
        """out: int64[2] <- (key, packed arc id) of the best candidate in shard `shard` of `shards`."""
        self._lib.emul_price(self._h, int(shard), int(shards), _ptr(out, ctypes.c_int64))

    def pivot(self, cands: np.ndarray, ncand: int) -> None:
        self._lib.emul_pivot(self._h, _ptr(cands, ctypes.c_int64), int(ncand))

    def pivots(self, cands: np.ndarray, ncand: int, count: int) -> None:
        self._lib.emul_pivots(self._h, _ptr(cands, ctypes.c_int64), int(ncand), int(count))

    def set_shards(self, shards: int) -> tuple[int, int]:
        """Size the (virtual) pricing grid for `shards` ranks; returns (list length per rank, minor pivots per sweep)."""
        self._lib.emul_set_shards(self._h, int(shards))
        return int(self._lib.emul_list_len(self._h)), int(self._lib.emul_minor_cap(self._h))

    def price_list(self, shard: int, shards: int, out: np.ndarray) -> None:
        """out: int64[2 * list length] <- (key, packed arc id) per pricing workgroup of the shard."""
        self._lib.emul_price_list(self._h, int(shard), int(shards), _ptr(out, ctypes.c_int64))

    def set_max_pivots(self, cap: int) -> None:
        self._lib.emul_set_max_pivots(self._h, int(cap))

    def poll(self, want_flow: bool = False):
        st, pv = ctypes.c_int32(0), ctypes.c_int64(0)
        obj = np.zeros(2, np.int64)
        flow = np.zeros(max(self.m, 1), np.int64) if want_flow else None
        self._lib.emul_poll(self._h, ctypes.byref(st), ctypes.byref(pv), _ptr(obj, ctypes.c_int64),
                            _ptr(flow, ctypes.c_int64) if want_flow else None)
        objective = (int(obj[0]) << 64) + (int(obj[1]) & ((1 << 64) - 1))
        return (None if st.value < 0 else int(st.value)), int(pv.value), objective, (flow[: self.m] if want_flow else None)

    def close(self):
        if self._h:
            self._lib.emul_destroy(self._h)
            self._h = None
