"""``NetworkSimplex`` facade: the reference's solver object, backed by the HIP engine.

Constructor / ``solve`` signature and the shape of everything that comes back
follow /root/reference/src/network_solver/simplex.py (``NetworkSimplex`` :62,
``__init__`` :99, ``solve`` :1446-1452).  What happens in between is different:

* the problem is flattened once into integer structure-of-arrays form
  (``flatten_problem``: the reference's node/arc ordering :149,:395, lower-bound
  shift :403-428, undirected expansion data.py:162-223, plus a decimal scaling
  step because the engine is integer);
* the pivot loop (:1109-1160, :1176-1425) runs on the MI355X through
  ``engine.McfEngine`` -- nothing in this module prices an arc or walks a tree;
* results are mapped back with the reference's post-processing (:1703-1765):
  flows summed per ``(tail, head)`` key, ``|f| <= tolerance`` dropped,
  ``round(., 12)``, objective over the original costs.

There is no CPU fallback here: without the HIP library or a GPU,
``NetworkSimplex(...)`` raises ``EngineUnavailableError``.
"""

from __future__ import annotations

import logging
import math
import time
from dataclasses import dataclass

import numpy as np

from . import engine as _engine
from .data import (ArrayBasis, Basis, FlowResult, LazyDuals, LazyFlows, NetworkProblem, ProgressCallback, ProgressInfo,
                   SoAProblem, SolverOptions)
from .exceptions import InvalidProblemError, SolverConfigurationError, UnboundedProblemError
from .specializations import analyze_network_structure, entering_rule_options

_MAX_DECIMALS = 9


@dataclass
class FlatProblem:
    """Integer SoA image of a NetworkProblem plus what is needed to map results back."""

    node_ids: list[str]            # index -> id, string-sorted like the reference
    keys: list[tuple[str, str]]    # per arc (tail id, head id), reference arc order
    tail: np.ndarray               # int32[m]
    head: np.ndarray               # int32[m]
    cost: np.ndarray               # int64[m]   cost * cost_scale
    cap: np.ndarray                # int64[m]   (capacity - lower) * flow_scale, -1 = unlimited
    supply: np.ndarray             # int64[n]   (supply shifted by lower bounds) * flow_scale
    lower: np.ndarray              # float64[m] original lower bounds (the shift)
    orig_cost: np.ndarray          # float64[m]
    flow_scale: int
    cost_scale: int
    soa: bool = False              # built from an SoAProblem: ids / keys are virtual sequences, results stay flat


class _IdSeq:
    """node index -> DIMACS id string, without holding n strings."""

    def __init__(self, n: int):
        self._n = n

    def __len__(self) -> int:
        return self._n

    def __getitem__(self, i: int) -> str:
        if not 0 <= i < self._n:
            raise IndexError(i)
        return str(i + 1)

    def __iter__(self):
        return (str(i + 1) for i in range(self._n))


class _KeySeq:
    """arc index -> (tail id, head id)."""

    def __init__(self, tail: np.ndarray, head: np.ndarray):
        self._tail, self._head = tail, head

    def __len__(self) -> int:
        return int(self._tail.shape[0])

    def __getitem__(self, i: int) -> tuple[str, str]:
        return (str(int(self._tail[i]) + 1), str(int(self._head[i]) + 1))


def flatten_soa(problem: SoAProblem, tolerance: float | None = None) -> FlatProblem:
    """SoAProblem -> the engine's arrays: the lower-bound shift of simplex.py:413-428 and the checks of :381-412,
    vectorised; no per-arc Python object.  Arcs and nodes keep the file's order (the reference's string sort of ids,
    :149 and :395, only decides among alternative optima and the order of pivots)."""
    tol = problem.tolerance if tolerance is None else tolerance
    m = problem.m
    lower = problem.lower
    finite = problem.capacity >= 0
    cap = np.full(m, -1, dtype=np.int64)
    width = problem.capacity - lower
    if (finite & (width < 0)).any():
        i = int(np.nonzero(finite & (width < 0))[0][0])
        raise InvalidProblemError(
            f"Arc capacity ({problem.capacity[i]}) is less than lower bound ({lower[i]}) for arc "
            f"{int(problem.tail[i]) + 1} -> {int(problem.head[i]) + 1}. Capacity must be >= lower bound.")
    cap[finite] = width[finite]
    supply = problem.supply.copy()
    if lower.any():
        np.subtract.at(supply, problem.tail, lower)
        np.add.at(supply, problem.head, lower)
    if abs(int(supply.sum())) > tol:
        raise InvalidProblemError(
            f"Supplies do not balance after lower-bound adjustment: total supply {float(supply.sum()):.6f} "
            f"exceeds tolerance {tol}.")
    if m and (np.abs(problem.cost).max() >= 2 ** 31 or cap.max() >= 2 ** 60):
        raise SolverConfigurationError("costs must fit int32 and capacities int60")
    return FlatProblem(_IdSeq(problem.n), _KeySeq(problem.tail, problem.head), problem.tail, problem.head, problem.cost,
                       cap, supply, lower.astype(np.float64), problem.cost.astype(np.float64), 1, 1, soa=True)


def _decimal_scale(values: np.ndarray, what: str) -> int:
    """Smallest power of ten that makes every value an integer (exactly, up to 1e-9 relative)."""
    if values.size == 0:
        return 1
    if not np.all(np.isfinite(values)):
        raise InvalidProblemError(f"{what} must be finite numbers")
    for k in range(_MAX_DECIMALS + 1):
        scaled = values * (10.0 ** k)
        if np.all(np.abs(scaled - np.round(scaled)) <= 1e-9 * np.maximum(1.0, np.abs(scaled))):
            return 10 ** k
    raise SolverConfigurationError(
        f"{what} need more than {_MAX_DECIMALS} decimal digits; the integer MI355X engine cannot "
        f"represent them exactly")


def flatten_problem(problem: NetworkProblem, tolerance: float | None = None) -> FlatProblem:
    """NetworkProblem -> integer SoA, in the reference's internal order.  ``tolerance`` is the solver's
    (``SolverOptions.tolerance``, simplex.py:154): the balance and bound checks below are the solver's own
    (:381-412), not the problem's build-time ones."""
    node_ids = sorted(problem.nodes.keys())                       # simplex.py:149
    index = {nid: i for i, nid in enumerate(node_ids)}
    arcs = sorted(problem.undirected_expansion(), key=lambda a: (a.tail, a.head))  # simplex.py:394-395
    m = len(arcs)
    tol = problem.tolerance if tolerance is None else tolerance
    supply = np.array([problem.nodes[nid].supply for nid in node_ids], dtype=np.float64)
    if abs(float(supply.sum())) > tol:                            # simplex.py:381-390
        raise InvalidProblemError(
            f"Supplies do not balance after lower-bound adjustment: total supply {supply.sum():.6f} "
            f"exceeds tolerance {tol}.")
    tail = np.fromiter((index[a.tail] for a in arcs), dtype=np.int32, count=m)
    head = np.fromiter((index[a.head] for a in arcs), dtype=np.int32, count=m)
    cost = np.fromiter((a.cost for a in arcs), dtype=np.float64, count=m)
    lower = np.fromiter((a.lower for a in arcs), dtype=np.float64, count=m)
    upper = np.empty(m, dtype=np.float64)
    for i, a in enumerate(arcs):                                  # simplex.py:399-412
        if a.capacity is None:
            upper[i] = math.inf
        else:
            u = float(a.capacity) - a.lower
            if u < -tol:
                raise InvalidProblemError(
                    f"Arc capacity ({a.capacity}) is less than lower bound ({a.lower}) for arc "
                    f"{a.tail} -> {a.head}. Capacity must be >= lower bound.")
            upper[i] = max(0.0, u)
    np.subtract.at(supply, tail, lower)                           # simplex.py:413-415
    np.add.at(supply, head, lower)
    finite = np.isfinite(upper)
    flow_scale = _decimal_scale(np.concatenate((supply, upper[finite], lower)), "supplies / capacities / lower bounds")
    cost_scale = _decimal_scale(cost, "costs")
    cap_i = np.full(m, -1, dtype=np.int64)
    cap_i[finite] = np.round(upper[finite] * flow_scale).astype(np.int64)
    supply_i = np.round(supply * flow_scale).astype(np.int64)
    residual = int(supply_i.sum())
    if residual != 0:
        # the reference tolerates |sum| <= tolerance (checked above); the integer engine needs an exact balance:
        # the sub-tolerance remainder goes onto the largest node -- never more than the tolerance allows
        if abs(residual) > tol * flow_scale + 1:
            raise InvalidProblemError(
                f"Supplies do not balance after lower-bound adjustment: total supply {residual / flow_scale:.6f} "
                f"exceeds tolerance {tol}.")
        supply_i[int(np.argmax(np.abs(supply_i)))] -= residual
    cost_i = np.round(cost * cost_scale).astype(np.int64)
    if m and (np.abs(cost_i).max() >= 2 ** 31 or (cap_i.max() >= 2 ** 60)):
        raise SolverConfigurationError("scaled costs must fit int32 and scaled capacities int60")
    return FlatProblem(node_ids, [(a.tail, a.head) for a in arcs], tail, head, cost_i, cap_i, supply_i, lower,
                       cost, flow_scale, cost_scale)


class NetworkSimplex:
    """Network simplex solver for minimum-cost flow, pivoting on an MI355X.

    Same construction and ``solve`` contract as the reference class
    (simplex.py:62-265, :1446-1765).  ``SolverOptions.pricing_strategy``:
    ``"dantzig"`` selects the full-scan pricing kernel, ``"devex"`` the block-search Devex
    kernel, ``"candidate_list"`` and ``"adaptive"`` (which the reference starts as a candidate
    list, simplex_pricing.py:545-587) the candidate-list rule: a full sweep keeps one candidate
    per pricing workgroup and the following pivots re-price only that list.  All rules reach the
    same optimum.
    """

    ROOT_NODE = "__network_simplex_root__"

    def __init__(self, problem: NetworkProblem, options: SolverOptions | None = None, *, device: int = -1,
                 batch_pivots: int = 64, use_graph: bool = True, engine_options: dict | None = None):
        self.options = options if options is not None else SolverOptions()
        self.logger = logging.getLogger(__name__)
        self.problem = problem
        self.tolerance = self.options.tolerance
        self.flat = (flatten_soa(problem, self.tolerance) if isinstance(problem, SoAProblem)
                     else flatten_problem(problem, self.tolerance))
        self.node_ids = [self.ROOT_NODE] + self.flat.node_ids if not self.flat.soa else self.flat.node_ids
        self.actual_arc_count = len(self.flat.keys)
        self.degenerate_pivots = 0
        strategy = self._select_pricing_strategy()
        self.pricing_rule = {"dantzig": _engine.RULE_DANTZIG, "devex": _engine.RULE_DEVEX_BLOCK}.get(
            strategy, _engine.RULE_CANDIDATE_LIST)  # candidate_list and adaptive (which starts as one)
        # simplex.py:133-137, 259-261 + specialized_pivots.py:452-527: structured instances get the reference's
        # specialised entering rule, as a variant of the same sweep kernel
        self.network_structure = analyze_network_structure(problem)
        special = entering_rule_options(self.network_structure, self.flat.node_ids, self.flat.tail, self.flat.head, self.flat.supply,
                                        unit=self.flat.flow_scale)
        if special is not None:
            # the reference tries the specialised scan first and falls back to the configured strategy when it finds nothing
            # (simplex.py:1060-1064); here the specialised rule IS a key variant of one sweep, so it replaces the strategy
            # for the whole solve -- status and objective are the same, iteration counts need not be
            if self.options.explicit_pricing_strategy and strategy != "dantzig":
                self.logger.info(f"pricing_strategy={strategy!r} is overridden by the specialised rule of this network class")
            self.pricing_rule = special.pop("rule")
            self.logger.info(f"Using specialized pivot strategy for {self.network_structure.network_type.value}")
        bs = self.options.block_size
        block_size = 0 if bs is None or isinstance(bs, str) else int(bs)
        self.engine = _engine.McfEngine(
            len(self.flat.node_ids), self.flat.tail, self.flat.head, self.flat.cost, self.flat.cap,
            self.flat.supply, rule=self.pricing_rule, block_size=block_size, batch_pivots=batch_pivots,
            use_graph=use_graph, device=device, **(special or {}), **(engine_options or {}))
        self.stats: dict = {}

    # simplex.py:314-374: the reference's grid-on-torus heuristic switches to Dantzig unless the
    # caller pinned a strategy
    def _select_pricing_strategy(self) -> str:
        if self.options.explicit_pricing_strategy:
            return self.options.pricing_strategy
        m = self.actual_arc_count
        if self.flat.soa:
            n = len(self.flat.node_ids)
            non_transship = int(np.count_nonzero(self.problem.supply))
        else:
            nodes = self.problem.nodes
            n = len(nodes)
            non_transship = sum(1 for nd in nodes.values() if abs(nd.supply) > self.tolerance)
        if n == 0:
            return self.options.pricing_strategy
        if (non_transship <= 4 and (n - non_transship) / n > 0.98 and (2 * m) / n >= 8 and 6 <= m / n <= 12):
            self.logger.info("Auto-detected grid-on-torus structure, switching to Dantzig pricing")
            return "dantzig"
        return self.options.pricing_strategy

    def _apply_warm_start_basis(self, basis: Basis) -> bool:
        """simplex.py:740-903: map the basis' (tail, head) keys onto arcs and hand them to the engine
        (``mcf_set_basis``), which adds one artificial arc per uncovered component and recomputes the tree
        flows from conservation (:905-1010).  False -> the engine is at the cold start (the reference's
        fall-back).  ``Basis.arc_flows`` entries of arcs OUTSIDE ``tree_arcs`` (this build's results carry the
        non-basic arcs that sit at capacity there; the reference ignores such keys) tell the engine which
        non-basic arcs start at their upper bound."""
        f = self.flat
        if isinstance(basis, ArrayBasis):                          # flat arrays in this problem's arc order
            if basis.in_tree.shape[0] != len(f.keys):
                self.logger.warning("Warm-start basis does not match the problem's arc count. Falling back to cold start.")
                return False
            if not basis.in_tree.any():
                self.logger.warning("Warm-start basis is empty. Falling back to cold start.")
                return False
            if self.engine.set_basis(basis.in_tree, basis.at_upper):
                self.logger.info(f"Successfully applied warm-start basis with {int(basis.in_tree.sum())} basis arcs")
                return True
            self.logger.warning(f"{self.engine.last_error()}. Falling back to cold start.")
            return False
        if len(basis.tree_arcs) == 0:
            self.logger.warning("Warm-start basis is empty. Falling back to cold start.")
            return False
        by_key: dict[tuple[str, str], list[int]] = {}
        for i, key in enumerate(f.keys):
            by_key.setdefault(key, []).append(i)
        in_tree = np.zeros(len(f.keys), dtype=np.int8)
        for key in basis.tree_arcs:
            idxs = by_key.get(tuple(key))
            if not idxs:
                self.logger.warning(f"Warm-start basis contains arc {key} not in current problem. "
                                    "Falling back to cold start.")
                return False
            in_tree[idxs[-1]] = 1                                  # like the reference's key -> index dict (:763-766)
        at_upper = np.zeros(len(f.keys), dtype=np.int8)
        for key, value in basis.arc_flows.items():
            idxs = by_key.get(tuple(key))
            if not idxs or tuple(key) in basis.tree_arcs:
                continue
            i = idxs[-1]
            if f.cap[i] > 0 and round(float(value) * f.flow_scale) >= f.cap[i]:
                at_upper[i] = 1
        if self.engine.set_basis(in_tree, at_upper):
            self.logger.info(f"Successfully applied warm-start basis with {int(in_tree.sum())} basis arcs")
            return True
        self.logger.warning(f"{self.engine.last_error()}. Falling back to cold start.")
        return False

    def _objective_estimate(self, flow: np.ndarray) -> float:
        f = self.flat
        return float(np.dot(flow / f.flow_scale + f.lower, f.orig_cost))

    def solve(self, max_iterations: int | None = None, progress_callback: ProgressCallback | None = None,
              progress_interval: int = 100, warm_start_basis: Basis | None = None) -> FlowResult:
        """Solve; returns a FlowResult, raises UnboundedProblemError (simplex.py:1446-1765)."""
        max_iterations = self.default_budget(max_iterations)    # simplex.py:1470 (len(arcs) incl. artificial)
        if warm_start_basis is not None:
            self.logger.info("Attempting to apply warm-start basis")      # simplex.py:1496
            if not self._apply_warm_start_basis(warm_start_basis):
                self.logger.info("Warm-start failed, performing cold start")  # simplex.py:1528
        start = time.time()

        progress = None
        raised: list[BaseException] = []
        if progress_callback is not None:
            def progress(pivots: int, cap: int, elapsed: float):
                res = self.engine.result()
                phase = 1 if res.stats["artificial_flow"] > 0 else 2
                try:
                    progress_callback(ProgressInfo(iteration=pivots, max_iterations=max_iterations, phase=phase,
                                                   phase_iterations=pivots,
                                                   objective_estimate=self._objective_estimate(res.flow),
                                                   elapsed_time=time.time() - start))
                except BaseException as exc:  # an exception cannot cross the C boundary: stop the solve, re-raise after
                    raised.append(exc)
                    return True
                return False

        self.engine.solve(max_iterations, progress, progress_interval)
        if raised:
            raise raised[0]
        return self._collect()

    def default_budget(self, max_iterations: int | None = None) -> int:
        """The pivot budget ``solve`` would use (simplex.py:1470)."""
        if max_iterations is None:
            max_iterations = self.options.max_iterations
        if max_iterations is None:
            max_iterations = max(100, 20 * (self.actual_arc_count + len(self.flat.node_ids)))
        return int(max_iterations)

    def _collect(self) -> FlowResult:
        """The engine's final state as the reference's FlowResult (simplex.py:1573-1765); raises UnboundedProblemError."""
        f = self.flat
        m = self.actual_arc_count
        res = self.engine.result()
        self.stats = res.stats
        iterations = int(res.stats["pivots"])
        self.degenerate_pivots = int(res.stats["degenerate"])

        if res.status == "unbounded":                             # simplex.py:1231-1246
            arc = int(res.stats["unbounded_arc"])
            raise UnboundedProblemError(
                "Unbounded problem detected: entering arc can increase indefinitely without hitting any "
                "capacity constraint. This indicates a negative-cost cycle with infinite capacity.",
                entering_arc=f.keys[arc] if 0 <= arc < m else None,
                reduced_cost=res.stats["unbounded_rc"] / f.cost_scale)
        if res.status == "infeasible" or (res.status == "iteration_limit" and res.stats["artificial_flow"] > 0):
            # simplex.py:1600-1624: no feasible flow (or none found within the budget)
            return FlowResult(objective=0.0, flows={}, status=res.status, iterations=iterations, duals={})

        if f.soa:
            # flat result: nothing per arc is boxed unless the caller looks at the dict views (simplex.py:1703-1765)
            flow_total = res.flow + self.problem.lower                       # flow + shift, exact integers
            objective = res.objective + int(np.dot(self.problem.lower.astype(object), self.problem.cost.astype(object))
                                            if self.problem.lower.any() else 0)
            at_upper = ~res.in_tree & (f.cap > 0) & (res.flow == f.cap)
            return FlowResult(objective=float(round(float(objective), 12)),
                              flows=LazyFlows(f.tail, f.head, flow_total, self.tolerance), status=res.status,
                              iterations=iterations, duals=LazyDuals(res.potential.astype(np.float64)),
                              basis=ArrayBasis(f.tail, f.head, res.in_tree, at_upper, res.flow))
        flow_value = res.flow.astype(np.float64) / f.flow_scale + f.lower   # flow + shift
        flows: dict[tuple[str, str], float] = {}
        objective = 0.0
        for i, key in enumerate(f.keys):                          # simplex.py:1703-1714
            fv = float(flow_value[i])
            flows[key] = flows.get(key, 0.0) + fv
            objective += fv * float(f.orig_cost[i])
        for key, value in list(flows.items()):                    # simplex.py:1716-1721
            if abs(value) <= self.tolerance:
                flows.pop(key)
            else:
                flows[key] = float(round(value, 12))
        duals = {nid: float(round(int(res.potential[i]) / f.cost_scale, 12)) for i, nid in enumerate(f.node_ids)}
        # simplex.py:1029-1039 (tree arcs with their flows), plus -- outside tree_arcs, where the reference
        # never looks -- the non-basic arcs sitting at capacity, so that a warm start can restore them
        tree_idx = np.nonzero(res.in_tree)[0]
        full_idx = np.nonzero(~res.in_tree & (f.cap > 0) & (res.flow == f.cap))[0]
        arc_flows = {f.keys[i]: float(res.flow[i]) / f.flow_scale for i in tree_idx}
        for i in full_idx:
            arc_flows.setdefault(f.keys[i], float(res.flow[i]) / f.flow_scale)
        basis = Basis(tree_arcs={f.keys[i] for i in tree_idx}, arc_flows=arc_flows)
        return FlowResult(objective=float(round(objective, 12)), flows=flows, status=res.status,
                          iterations=iterations, duals=duals, basis=basis)
