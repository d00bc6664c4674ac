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
from collections.abc import Callable, Iterable, Sequence
from dataclasses import dataclass, field

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
