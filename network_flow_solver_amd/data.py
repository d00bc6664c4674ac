"""Boundary types: the reference's problem / option / result objects, re-stated.

Field names, defaults and validation behaviour follow
/root/reference/src/network_solver/data.py so that code written against the
reference (``build_problem``, ``SolverOptions(...)``, ``result.flows[(u, v)]``)
runs unchanged on the MI355X engine.  Only the knobs that steer the dense-basis
machinery (Forrest-Tomlin / LU / condition number) are inert here: they are
accepted and validated, and then ignored, because a spanning-tree basis needs
no factorisation.
"""

from __future__ import annotations

import math
from collections.abc import Callable, Iterable, Mapping, Sequence
from dataclasses import dataclass, field

import numpy as np

from .exceptions import InvalidProblemError

PRICING_STRATEGIES = ("devex", "dantzig", "candidate_list", "adaptive")


@dataclass(frozen=True)
class Node:
    """Network node; ``supply`` > 0 produces, < 0 consumes (data.py:12-37)."""

    id: str
    supply: float = 0.0


@dataclass(frozen=True)
class Arc:
    """Directed arc with capacity (None = unlimited), unit cost and lower bound (data.py:40-88)."""

    tail: str
    head: str
    capacity: float | None
    cost: float
    lower: float = 0.0

    def __post_init__(self) -> None:
        if self.tail == self.head:
            raise InvalidProblemError(
                f"Self-loop detected on node '{self.tail}'. Self-loops are not supported in network simplex.")
        if self.capacity is not None and self.capacity < self.lower:
            raise InvalidProblemError(
                f"Arc {self.tail} -> {self.head} has capacity ({self.capacity}) less than lower bound "
                f"({self.lower}). Capacity must be >= lower bound.")


@dataclass
class NetworkProblem:
    """A minimum-cost flow instance (data.py:91-223)."""

    directed: bool
    nodes: dict[str, Node]
    arcs: list[Arc]
    tolerance: float = 1e-3

    def validate(self) -> None:
        total = sum(node.supply for node in self.nodes.values())
        if abs(total) > self.tolerance:
            raise InvalidProblemError(
                f"Problem is unbalanced: total supply {total:.6f} exceeds tolerance {self.tolerance}. "
                f"The sum of all node supplies must equal zero.")
        for arc in self.arcs:
            for end, label in ((arc.tail, "tail"), (arc.head, "head")):
                if end not in self.nodes:
                    raise InvalidProblemError(
                        f"Arc {label} '{end}' not found in node set. All arc endpoints must reference "
                        f"existing nodes.")

    def undirected_expansion(self) -> Sequence[Arc]:
        """Undirected edge {u, v} with capacity C -> arc (u, v) with bounds [-C, C] (data.py:162-223)."""
        if self.directed:
            return tuple(self.arcs)
        out: list[Arc] = []
        for arc in self.arcs:
            if arc.capacity is None:
                raise InvalidProblemError(
                    f"Undirected edge {arc.tail} -- {arc.head} has infinite capacity. Undirected graphs "
                    f"require finite capacity on all edges.")
            cap = float(arc.capacity)
            if abs(arc.lower) > 1e-12 and not math.isclose(arc.lower, -cap, rel_tol=0.0, abs_tol=1e-12):
                raise InvalidProblemError(
                    f"Undirected edge {arc.tail} -- {arc.head} has custom lower bound ({arc.lower}). "
                    f"Undirected edges do not support custom lower bounds.")
            out.append(Arc(tail=arc.tail, head=arc.head, capacity=cap, cost=arc.cost, lower=-cap))
        return tuple(out)


class SoAProblem:
    """A DIRECTED minimum-cost flow instance in flat integer arrays -- what the native DIMACS reader returns and the
    engine consumes -- behind the ``NetworkProblem`` surface (``directed``, ``tolerance``, ``nodes``, ``arcs``,
    ``validate``, ``undirected_expansion``).  Node ids are the DIMACS strings "1" .. "n" (node index + 1).

    Purpose: the reference's loader builds one ``Node`` / ``Arc`` dataclass per element
    (benchmarks/parsers/dimacs.py:77-102 -> data.py:532-567), which is what makes a 16 M-arc file unloadable there.
    ``solve_min_cost_flow`` / ``NetworkSimplex`` accept this object directly and never touch ``nodes`` / ``arcs``;
    those two are materialised lazily for code that really wants the object model (O(n + m) Python objects)."""

    directed = True

    def __init__(self, n: int, tail, head, cost, capacity, supply, lower=None, tolerance: float = 1e-6,
                 name: str = "instance"):
        self.n = int(n)
        self.tail = np.ascontiguousarray(tail, dtype=np.int32)
        self.head = np.ascontiguousarray(head, dtype=np.int32)
        self.cost = np.ascontiguousarray(cost, dtype=np.int64)
        self.capacity = np.ascontiguousarray(capacity, dtype=np.int64)        # -1 = unlimited
        self.supply = np.ascontiguousarray(supply, dtype=np.int64)
        m = self.tail.shape[0]
        self.lower = np.zeros(m, dtype=np.int64) if lower is None else np.ascontiguousarray(lower, dtype=np.int64)
        self.tolerance = float(tolerance)
        self.name = name
        self._nodes = None
        self._arcs = None
        self.validate()

    @property
    def m(self) -> int:
        return int(self.tail.shape[0])

    @property
    def cap(self) -> np.ndarray:   # the generators' name for the same array
        return self.capacity

    def validate(self) -> None:
        """The checks of NetworkProblem.validate and Arc.__post_init__ (data.py:78-88, 141-160), vectorised."""
        m = self.m
        if not (self.head.shape[0] == self.cost.shape[0] == self.capacity.shape[0] == self.lower.shape[0] == m):
            raise InvalidProblemError("Arc arrays differ in length.")
        if self.supply.shape[0] != self.n:
            raise InvalidProblemError("Supply array must have one entry per node.")
        total = int(self.supply.sum())
        if abs(total) > self.tolerance:
            raise InvalidProblemError(
                f"Problem is unbalanced: total supply {float(total):.6f} exceeds tolerance {self.tolerance}. "
                f"The sum of all node supplies must equal zero.")
        if m == 0:
            return
        bad = (self.tail < 0) | (self.tail >= self.n) | (self.head < 0) | (self.head >= self.n)
        if bad.any():
            i = int(np.nonzero(bad)[0][0])
            raise InvalidProblemError(
                f"Arc tail '{int(self.tail[i]) + 1}' or head '{int(self.head[i]) + 1}' not found in node set. "
                f"All arc endpoints must reference existing nodes.")
        loops = self.tail == self.head
        if loops.any():
            i = int(np.nonzero(loops)[0][0])
            raise InvalidProblemError(
                f"Self-loop detected on node '{int(self.tail[i]) + 1}'. Self-loops are not supported in network simplex.")
        short = (self.capacity >= 0) & (self.capacity < self.lower)
        if short.any():
            i = int(np.nonzero(short)[0][0])
            raise InvalidProblemError(
                f"Arc {int(self.tail[i]) + 1} -> {int(self.head[i]) + 1} has capacity ({float(self.capacity[i])}) less than "
                f"lower bound ({float(self.lower[i])}). Capacity must be >= lower bound.")

    # ---- the object model, on demand
    @property
    def nodes(self) -> dict[str, Node]:
        if self._nodes is None:
            self._nodes = {str(i + 1): Node(id=str(i + 1), supply=float(s)) for i, s in enumerate(self.supply.tolist())}
        return self._nodes

    @property
    def arcs(self) -> list[Arc]:
        if self._arcs is None:
            self._arcs = [Arc(tail=str(t + 1), head=str(h + 1), capacity=None if cp < 0 else float(cp), cost=float(c),
                              lower=float(lo))
                          for t, h, cp, c, lo in zip(self.tail.tolist(), self.head.tolist(), self.capacity.tolist(),
                                                     self.cost.tolist(), self.lower.tolist())]
        return self._arcs

    def undirected_expansion(self) -> Sequence[Arc]:
        return tuple(self.arcs)

    def to_network_problem(self) -> NetworkProblem:
        return NetworkProblem(directed=True, nodes=dict(self.nodes), arcs=list(self.arcs), tolerance=self.tolerance)


class LazyFlows(Mapping):
    """``FlowResult.flows`` of an SoA solve: the reference's dict view -- parallel arcs summed per ``(tail, head)`` key,
    ``|f| <= tolerance`` dropped, ``round(f, 12)`` (simplex.py:1703-1721) -- built from the flat flow array on first
    use.  ``array`` is the per-arc flow in the problem's arc order."""

    def __init__(self, tail: np.ndarray, head: np.ndarray, flow: np.ndarray, tolerance: float):
        self._tail, self._head, self.array, self._tol = tail, head, flow, tolerance
        self._dict: dict[tuple[str, str], float] | None = None

    def _build(self) -> dict[tuple[str, str], float]:
        if self._dict is None:
            nz = np.nonzero(self.array)[0]
            out: dict[tuple[str, str], float] = {}
            for t, h, f in zip((self._tail[nz] + 1).tolist(), (self._head[nz] + 1).tolist(), self.array[nz].tolist()):
                key = (str(t), str(h))
                out[key] = out.get(key, 0.0) + float(f)
            self._dict = {k: float(round(v, 12)) for k, v in out.items() if abs(v) > self._tol}
        return self._dict

    def __getitem__(self, key):
        return self._build()[key]

    def __iter__(self):
        return iter(self._build())

    def __len__(self) -> int:
        return len(self._build())

    def __repr__(self) -> str:
        return f"LazyFlows({int(np.count_nonzero(self.array))} non-zero arc flows)"


class LazyDuals(Mapping):
    """``FlowResult.duals`` of an SoA solve: ``{"i": round(pi, 12)}`` built on first use; ``array`` holds them flat."""

    def __init__(self, values: np.ndarray):
        self.array = values
        self._dict: dict[str, float] | None = None

    def _build(self) -> dict[str, float]:
        if self._dict is None:
            self._dict = {str(i + 1): float(round(v, 12)) for i, v in enumerate(self.array.tolist())}
        return self._dict

    def __getitem__(self, key):
        return self._build()[key]

    def __iter__(self):
        return iter(self._build())

    def __len__(self) -> int:
        return int(self.array.shape[0])


class ArrayBasis:
    """Warm-start basis of an SoA solve: what ``Basis`` holds (data.py:226-266), as per-arc arrays in the problem's
    arc order.  ``tree_arcs`` / ``arc_flows`` give the reference's set / dict view on demand."""

    def __init__(self, tail: np.ndarray, head: np.ndarray, in_tree: np.ndarray, at_upper: np.ndarray, flow: np.ndarray):
        self._tail, self._head = tail, head
        self.in_tree = np.ascontiguousarray(in_tree, dtype=np.int8)
        self.at_upper = np.ascontiguousarray(at_upper, dtype=np.int8)
        self.flow = flow

    def _keys(self, idx):
        return [(str(t), str(h)) for t, h in zip((self._tail[idx] + 1).tolist(), (self._head[idx] + 1).tolist())]

    @property
    def tree_arcs(self) -> set[tuple[str, str]]:
        return set(self._keys(np.nonzero(self.in_tree)[0]))

    @property
    def arc_flows(self) -> dict[tuple[str, str], float]:
        idx = np.nonzero(self.in_tree | self.at_upper)[0]
        return dict(zip(self._keys(idx), (float(f) for f in self.flow[idx].tolist())))


@dataclass
class Basis:
    """Spanning-tree basis for warm starts (data.py:226-266)."""

    tree_arcs: set[tuple[str, str]] = field(default_factory=set)
    arc_flows: dict[tuple[str, str], float] = field(default_factory=dict)


@dataclass
class FlowResult:
    """Solver output (data.py:269-322). ``status`` is one of optimal / infeasible / iteration_limit."""

    objective: float
    flows: dict[tuple[str, str], float] = field(default_factory=dict)
    status: str = "optimal"
    iterations: int = 0
    duals: dict[str, float] = field(default_factory=dict)
    basis: Basis | None = None


@dataclass(frozen=True)
class ProgressInfo:
    """Payload of the progress callback (data.py:325-343)."""

    iteration: int
    max_iterations: int
    phase: int
    phase_iterations: int
    objective_estimate: float
    elapsed_time: float


ProgressCallback = Callable[[ProgressInfo], None]


@dataclass
class SolverOptions:
    """Solver configuration (data.py:350-529): same fields, defaults and validation errors."""

    max_iterations: int | None = None
    tolerance: float = 1e-6
    pricing_strategy: str = "adaptive"
    explicit_pricing_strategy: bool = False
    block_size: int | str | None = None
    ft_update_limit: int = 64
    projection_cache_size: int = 100
    auto_scale: bool = True
    adaptive_refactorization: bool = True
    condition_check_interval: int = 50
    condition_number_threshold: float = 1e12
    adaptive_ft_min: int = 20
    adaptive_ft_max: int = 200
    use_dense_inverse: bool | None = None
    use_vectorized_pricing: bool = True
    use_jit: bool = True

    def __post_init__(self) -> None:
        if self.tolerance <= 0:
            raise InvalidProblemError(f"Tolerance must be positive, got {self.tolerance}.")
        if self.pricing_strategy not in PRICING_STRATEGIES:
            raise InvalidProblemError(
                f"Invalid pricing strategy '{self.pricing_strategy}'. Must be 'devex', 'dantzig', "
                f"'candidate_list', or 'adaptive'.")
        if self.block_size is not None:
            if isinstance(self.block_size, str):
                if self.block_size != "auto":
                    raise InvalidProblemError(
                        f"Invalid block_size '{self.block_size}'. Must be a positive integer, 'auto', or None.")
            elif self.block_size <= 0:
                raise InvalidProblemError(f"Block size must be positive, got {self.block_size}.")
        if self.ft_update_limit <= 0:
            raise InvalidProblemError(f"FT update limit must be positive, got {self.ft_update_limit}.")
        if self.condition_number_threshold <= 1:
            raise InvalidProblemError(
                f"Condition number threshold must be > 1, got {self.condition_number_threshold}.")
        if self.adaptive_ft_min <= 0 or self.adaptive_ft_min > self.adaptive_ft_max:
            raise InvalidProblemError(
                f"Adaptive FT min must be positive and <= max, got min={self.adaptive_ft_min}, "
                f"max={self.adaptive_ft_max}.")
        if self.use_dense_inverse is None:
            # the reference resolves None by probing for scipy (data.py:512-517); the tree basis of
            # this engine never builds an inverse, so the resolved value is simply False
            object.__setattr__(self, "use_dense_inverse", False)


def build_problem(nodes: Iterable[dict], arcs: Iterable[dict], directed: bool, tolerance: float) -> NetworkProblem:
    """Assemble and validate a NetworkProblem from plain dictionaries (data.py:532-567)."""
    node_map: dict[str, Node] = {}
    for nd in nodes:
        node_id = str(nd["id"])
        if node_id in node_map:
            raise InvalidProblemError(f"Duplicate node id '{node_id}'. Each node must have a unique identifier.")
        node_map[node_id] = Node(id=node_id, supply=float(nd.get("supply", 0.0)))
    arc_list: list[Arc] = []
    for a in arcs:
        cap = a.get("capacity")
        arc_list.append(Arc(tail=str(a["tail"]), head=str(a["head"]),
                            capacity=None if cap is None else float(cap),
                            cost=float(a.get("cost", 0.0)), lower=float(a.get("lower", 0.0))))
    problem = NetworkProblem(directed=directed, nodes=node_map, arcs=arc_list, tolerance=float(tolerance))
    problem.validate()
    return problem
