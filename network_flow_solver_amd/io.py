"""Problem and result files in the reference's JSON schema.

Schema (reference ``io.py:33-69``): a problem is an object with ``nodes`` (``[{id, supply}]``), ``edges`` or
``arcs`` (``[{tail, head, capacity?, cost?, lower?}]``), optional ``directed`` (default true) and ``tolerance``
(default 1e-3); a result file holds ``status, objective, iterations, flows[{tail, head, flow}], duals{}`` with
flows and duals in key order.
"""

from __future__ import annotations

import json
import os

from .data import FlowResult, NetworkProblem, build_problem
from .exceptions import InvalidProblemError

_ARC_DEFAULTS = (("capacity", None), ("cost", 0.0), ("lower", 0.0))


def _arc_record(entry) -> dict:
    """One arc entry with the optional fields filled in; end points are mandatory."""
    try:
        record = {"tail": entry["tail"], "head": entry["head"]}
    except (KeyError, TypeError):
        raise InvalidProblemError(
            f"Invalid edge specification: {entry}. Each edge must have 'tail' and 'head' fields.") from None
    for field, default in _ARC_DEFAULTS:
        record[field] = entry.get(field, default)
    return record


def load_problem(path: str | os.PathLike) -> NetworkProblem:
    """Parse a JSON problem file into a ``NetworkProblem`` (validation happens in ``build_problem``)."""
    with open(path, encoding="utf-8") as handle:
        doc = json.load(handle)
    node_list = doc.get("nodes")
    arc_list = doc.get("edges") or doc.get("arcs")
    if not (isinstance(node_list, list) and isinstance(arc_list, list)):
        found_arcs = type(arc_list).__name__ if arc_list else "None"
        raise InvalidProblemError(
            "Invalid problem format: JSON must include 'nodes' and 'edges' (or 'arcs') arrays. "
            f"Got nodes type: {type(node_list).__name__}, edges type: {found_arcs}")
    return build_problem(nodes=node_list, arcs=[_arc_record(a) for a in arc_list],
                         directed=bool(doc.get("directed", True)), tolerance=float(doc.get("tolerance", 1e-3)))


def save_result(path: str | os.PathLike, result: FlowResult) -> None:
    """Write a ``FlowResult`` as JSON, flows and duals ordered by key so files diff cleanly."""
    flow_rows = []
    for key in sorted(result.flows):
        flow_rows.append({"tail": key[0], "head": key[1], "flow": result.flows[key]})
    document = {"status": result.status, "objective": result.objective, "iterations": result.iterations,
                "flows": flow_rows, "duals": {node: result.duals[node] for node in sorted(result.duals)}}
    with open(path, "w", encoding="utf-8") as handle:
        json.dump(document, handle, indent=2)
