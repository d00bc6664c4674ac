"""JSON problem / result files in the reference's schema (/root/reference/src/network_solver/io.py)."""

from __future__ import annotations

import json
from collections.abc import Iterable, Mapping
from pathlib import Path
from typing import Any

from .data import FlowResult, NetworkProblem, build_problem
from .exceptions import InvalidProblemError


def _normalize_edges(raw: Iterable[Mapping[str, Any]]) -> list[dict[str, Any]]:
    """Uniform edge dictionaries; ``tail`` and ``head`` are mandatory (io.py:12-30)."""
    out = []
    for edge in raw:
        if "tail" not in edge or "head" not in edge:
            raise InvalidProblemError(
                f"Invalid edge specification: {edge}. Each edge must have 'tail' and 'head' fields.")
        out.append({"tail": edge["tail"], "head": edge["head"], "capacity": edge.get("capacity"),
                    "cost": edge.get("cost", 0.0), "lower": edge.get("lower", 0.0)})
    return out


def load_problem(path: str | Path) -> NetworkProblem:
    """Read ``{directed, tolerance, nodes[], edges[] | arcs[]}`` (io.py:33-52)."""
    with Path(path).open("r", encoding="utf-8") as fh:
        payload = json.load(fh)
    nodes = payload.get("nodes")
    edges = payload.get("edges") or payload.get("arcs")
    if not isinstance(nodes, list) or not isinstance(edges, list):
        raise InvalidProblemError(
            "Invalid problem format: JSON must include 'nodes' and 'edges' (or 'arcs') arrays. "
            f"Got nodes type: {type(nodes).__name__}, edges type: {type(edges).__name__ if edges else 'None'}")
    return build_problem(nodes=nodes, arcs=_normalize_edges(edges), directed=bool(payload.get("directed", True)),
                         tolerance=float(payload.get("tolerance", 1e-3)))


def save_result(path: str | Path, result: FlowResult) -> None:
    """Write status / objective / iterations / flows (sorted by key) / duals (io.py:55-69)."""
    data = {
        "status": result.status,
        "objective": result.objective,
        "iterations": result.iterations,
        "flows": [{"tail": t, "head": h, "flow": f} for (t, h), f in sorted(result.flows.items())],
        "duals": dict(sorted(result.duals.items())),
    }
    with Path(path).open("w", encoding="utf-8") as fh:
        json.dump(data, fh, indent=2, sort_keys=False)
