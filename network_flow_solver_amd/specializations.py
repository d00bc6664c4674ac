"""Network structure analysis: which special problem class an instance belongs to.

Same surface as /root/reference/src/network_solver/specializations.py (``NetworkType``,
``NetworkStructure``, ``analyze_network_structure``) because the class decides which
specialised entering rule the solver tries first (specialized_pivots.py:452-527, dispatch
simplex.py:1061-1064).  The implementation is array based -- supplies and arc end points as
numpy vectors, the 2-colouring from one scipy breadth-first pass per component -- so that it
costs O(n + m) vector work on a 16 M-arc ``SoAProblem`` instead of one Python step per arc.

How the classes map onto the MI355X engine's pricing rules (``NetworkSimplex``):

=====================  ==========================================================================
TRANSPORTATION         row scan = most violating arc in either direction, first index on ties
                       (specialized_pivots.py:69-117): exactly the full-scan Dantzig kernel
ASSIGNMENT             min-cost scan over forward arcs only, the general rule for what is left
                       (:191-223): the Dantzig kernel with ``forward_first`` keys
SHORTEST_PATH          label-gated Dantzig (:338-424): forward arcs whose tail the source reaches and every
                       backward arc first, the general rule for the rest: ``key_mode = KEY_PRIORITY``
BIPARTITE_MATCHING     arcs leaving the unit-supply nodes of the left partition first (:233-281; the engine
                       only ever enters improving arcs): ``KEY_PRIORITY``
MAX_FLOW               merit = residual x |reduced cost| (:284-335): ``KEY_CAPACITY``
=====================  ==========================================================================
"""

from __future__ import annotations

from dataclasses import dataclass, field
from enum import Enum

import numpy as np

from .data import SoAProblem


class NetworkType(Enum):
    GENERAL = "general"
    TRANSPORTATION = "transportation"
    ASSIGNMENT = "assignment"
    BIPARTITE_MATCHING = "bipartite_matching"
    MAX_FLOW = "max_flow"
    SHORTEST_PATH = "shortest_path"


@dataclass
class NetworkStructure:
    network_type: NetworkType
    is_bipartite: bool
    source_nodes: set = field(default_factory=set)
    sink_nodes: set = field(default_factory=set)
    transshipment_nodes: set = field(default_factory=set)
    partitions: tuple | None = None
    total_supply: float = 0.0
    total_demand: float = 0.0
    is_balanced: bool = True
    has_lower_bounds: bool = False
    has_finite_capacities: bool = True


def _two_colouring(n: int, tail: np.ndarray, head: np.ndarray):
    """colour[n] in {0, 1} of a proper 2-colouring of the undirected arc graph (every component started at its
    lowest node with colour 0), or None when an odd cycle exists."""
    if n == 0:
        return None
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import breadth_first_order, connected_components

    ones = np.ones(tail.shape[0], dtype=np.int8)
    graph = coo_matrix((ones, (tail, head)), shape=(n, n)).tocsr()
    ncomp, label = connected_components(graph, directed=False)
    depth = np.zeros(n, dtype=np.int64)
    first = np.full(ncomp, n, dtype=np.int64)
    np.minimum.at(first, label, np.arange(n))
    for root in first.tolist():
        order, pred = breadth_first_order(graph, int(root), directed=False, return_predecessors=True)
        for v in order[1:].tolist():                      # BFS order: the predecessor's depth is final
            depth[v] = depth[pred[v]] + 1
    colour = (depth & 1).astype(np.int8)
    if tail.shape[0] and (colour[tail] == colour[head]).any():
        return None
    return colour


def _flat_view(problem):
    """(ids, supply[n], tail[m], head[m], lower[m], capacity-is-finite[m], cost[m]) whatever the problem class."""
    if isinstance(problem, SoAProblem):
        ids = None
        return (ids, problem.supply.astype(np.float64), problem.tail, problem.head, problem.lower.astype(np.float64),
                problem.capacity >= 0, problem.cost.astype(np.float64))
    ids = list(problem.nodes.keys())
    index = {nid: i for i, nid in enumerate(ids)}
    supply = np.array([problem.nodes[i].supply for i in ids], dtype=np.float64)
    m = len(problem.arcs)
    tail = np.fromiter((index[a.tail] for a in problem.arcs), dtype=np.int64, count=m)
    head = np.fromiter((index[a.head] for a in problem.arcs), dtype=np.int64, count=m)
    lower = np.fromiter((a.lower for a in problem.arcs), dtype=np.float64, count=m)
    finite = np.fromiter((a.capacity is not None and a.capacity < float("inf") for a in problem.arcs), dtype=bool, count=m)
    cost = np.fromiter((a.cost for a in problem.arcs), dtype=np.float64, count=m)
    return ids, supply, tail, head, lower, finite, cost


def analyze_network_structure(problem) -> NetworkStructure:
    """Classify a NetworkProblem / SoAProblem (specializations.py:60-288: node roles by the problem's tolerance,
    bipartiteness of the undirected arc graph, then the tests in the reference's order)."""
    tol = problem.tolerance
    ids, supply, tail, head, lower, finite, cost = _flat_view(problem)
    n = supply.shape[0]
    name = (lambda i: str(i + 1)) if ids is None else (lambda i: ids[i])
    src = supply > tol
    snk = supply < -tol
    mid = ~(src | snk)
    total_supply = float(supply[src].sum())
    total_demand = float(-supply[snk].sum())
    balanced = abs(total_supply - total_demand) <= tol
    has_lower = bool((lower > tol).any())
    ns, nk, nt = int(src.sum()), int(snk.sum()), int(mid.sum())
    unit_values = bool(((np.abs(np.abs(supply) - 1.0) <= tol) | (np.abs(supply) <= tol)).all())
    # the 2-colouring is only ever consulted when one of these holds: skip it for the general DIMACS families
    need_colouring = n > 0 and (n <= 100_000 or (not has_lower and ((nt == 0 and ns > 0 and nk > 0) or unit_values)))
    colour = _two_colouring(n, tail, head) if need_colouring else None
    bipartite = colour is not None
    small = n <= 100_000                                   # id sets are for inspection; not materialised at scale

    def ids_of(mask):
        return {name(i) for i in np.nonzero(mask)[0].tolist()} if small else set()

    out = NetworkStructure(
        network_type=NetworkType.GENERAL, is_bipartite=bipartite, source_nodes=ids_of(src), sink_nodes=ids_of(snk),
        transshipment_nodes=ids_of(mid),
        partitions=(ids_of(colour == 0), ids_of(colour == 1)) if bipartite else None,
        total_supply=total_supply, total_demand=total_demand, is_balanced=balanced, has_lower_bounds=has_lower,
        has_finite_capacities=bool(finite.all()))
    kind = NetworkType.GENERAL
    if nt == 0 and ns > 0 and nk > 0 and bipartite and not has_lower and bool((src[tail] & snk[head]).all()):
        unit = balanced and ns == nk and bool((np.abs(supply[src] - 1.0) <= tol).all()) \
            and bool((np.abs(supply[snk] + 1.0) <= tol).all())
        kind = NetworkType.ASSIGNMENT if unit else NetworkType.TRANSPORTATION
    elif ns == 1 and nk == 1 and abs(float(supply[src][0]) - 1.0) <= tol and abs(float(supply[snk][0]) + 1.0) <= tol:
        kind = NetworkType.SHORTEST_PATH
    elif bipartite and not has_lower and unit_values:
        kind = NetworkType.BIPARTITE_MATCHING
    elif ns == 1 and nk == 1 and not has_lower and cost.shape[0] >= 0 and \
            (bool((np.abs(cost) <= tol).all()) or bool((np.abs(cost - 1.0) <= tol).all())):
        kind = NetworkType.MAX_FLOW
    out.network_type = kind
    return out


def entering_rule_options(structure: NetworkStructure, node_ids, tail: np.ndarray, head: np.ndarray, supply: np.ndarray,
                          unit: int = 1) -> dict | None:
    """Engine options (``McfEngine`` keyword arguments) of the specialised entering rule the reference would pick for this
    structure (specialized_pivots.py:452-527), or None for the general pricing rule.  ``tail`` / ``head`` / ``supply``
    are the solver's flat integer arrays (supplies in multiples of 1 / ``unit``), ``node_ids`` the ids in the same node order.  Every variant is a key of the
    Dantzig sweep (``mcf_core.h: mcf_dantzig_key``), so the rule is RULE_DANTZIG with a ``key_mode``."""
    from . import engine as _engine

    kind = structure.network_type
    n = int(supply.shape[0])
    if kind is NetworkType.TRANSPORTATION:                       # row scan == the plain full-scan Dantzig kernel
        return {"rule": _engine.RULE_DANTZIG}
    if kind is NetworkType.ASSIGNMENT:
        return {"rule": _engine.RULE_DANTZIG, "key_mode": _engine.KEY_FORWARD_FIRST}
    if kind is NetworkType.MAX_FLOW:                             # needs a source and a sink (:488-503)
        if (supply > 0).any() and (supply < 0).any():
            return {"rule": _engine.RULE_DANTZIG, "key_mode": _engine.KEY_CAPACITY}
        return None
    if kind is NetworkType.SHORTEST_PATH:                        # one unit from a source to a sink (:505-522)
        src = np.flatnonzero(supply == unit)
        if src.size == 0 or not (supply == -unit).any():
            return None
        from scipy.sparse import coo_matrix
        from scipy.sparse.csgraph import breadth_first_order

        graph = coo_matrix((np.ones(tail.shape[0], dtype=np.int8), (tail, head)), shape=(n, n)).tocsr()
        reached = np.zeros(n, dtype=bool)
        reached[breadth_first_order(graph, int(src[0]), directed=True, return_predecessors=False)] = True
        prio = np.where(reached[tail], 3, 2).astype(np.int8)     # bit 1: every backward candidate; bit 0: forward with a labelled tail
        return {"rule": _engine.RULE_DANTZIG, "key_mode": _engine.KEY_PRIORITY, "arc_priority": prio}
    if kind is NetworkType.BIPARTITE_MATCHING and structure.partitions is not None and structure.partitions[0]:
        left = structure.partitions[0]
        in_left = np.fromiter((nid in left for nid in node_ids), dtype=bool, count=n)
        prio = (in_left[tail] & (supply[tail] == unit)).astype(np.int8)
        return {"rule": _engine.RULE_DANTZIG, "key_mode": _engine.KEY_PRIORITY, "arc_priority": prio}
    return None
