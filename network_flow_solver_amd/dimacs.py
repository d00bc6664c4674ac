"""DIMACS ``p min`` loader with the reference parser's behaviour
(/root/reference/benchmarks/parsers/dimacs.py:46-286):

* ``c`` comments, one ``p min <nodes> <arcs>``, ``n <id> <supply>``,
  ``a <tail> <head> <lower> <cap> <cost>`` or the 4-field ``a <tail> <head> <cap> <cost>``;
* capacity ``-1`` / ``inf`` / >= 1e15 means unlimited (dimacs.py:216-221);
* node ids are the strings "1".."n"; arcs must stay in that range; tolerance 1e-6.

``parse_dimacs_soa`` is the native-layout variant: it goes straight to the flat
integer arrays the engine consumes, skipping the per-arc Python objects that
make the object path infeasible at 10^6+ arcs.
"""

from __future__ import annotations

from pathlib import Path

import numpy as np

from .data import NetworkProblem, build_problem
from .exceptions import InvalidProblemError
from .generators import ArcSoA


def _scan(lines):
    """Yield validated (kind, fields, line_no) records; shared by both front ends."""
    seen_p = False
    n_nodes = n_arcs = None
    supplies: dict[str, float] = {}
    arcs: list[tuple[str, str, float, float | None, float]] = []
    for line_no, raw in enumerate(lines, start=1):
        line = raw.strip()
        if not line or line.startswith("c"):
            continue
        tok = line.split()
        kind = tok[0]
        try:
            if kind == "p":
                if seen_p:
                    raise InvalidProblemError(
                        f"Line {line_no}: Multiple problem descriptor lines found. Only one 'p min' line is allowed.")
                if len(tok) != 4:
                    raise InvalidProblemError(
                        f"Line {line_no}: Invalid problem descriptor format. Expected 'p min <nodes> <arcs>', got: {line}")
                if tok[1] != "min":
                    raise InvalidProblemError(
                        f"Line {line_no}: Only 'min' (minimum cost flow) problems supported. Got: {tok[1]}")
                n_nodes, n_arcs = int(tok[2]), int(tok[3])
                seen_p = True
                if n_nodes <= 0:
                    raise InvalidProblemError(f"Line {line_no}: Number of nodes must be positive, got {n_nodes}")
                if n_arcs < 0:
                    raise InvalidProblemError(f"Line {line_no}: Number of arcs cannot be negative, got {n_arcs}")
            elif kind == "n":
                if not seen_p:
                    raise InvalidProblemError(
                        f"Line {line_no}: Node descriptor before problem descriptor. The 'p min' line must come first.")
                if len(tok) != 3:
                    raise InvalidProblemError(
                        f"Line {line_no}: Invalid node descriptor format. Expected 'n <node_id> <supply>', got: {line}")
                supplies[tok[1]] = float(tok[2])
            elif kind == "a":
                if not seen_p:
                    raise InvalidProblemError(
                        f"Line {line_no}: Arc descriptor before problem descriptor. The 'p min' line must come first.")
                if len(tok) == 6:
                    tail, head, lower, cap_s, cost = tok[1], tok[2], float(tok[3]), tok[4], float(tok[5])
                elif len(tok) == 5:
                    tail, head, lower, cap_s, cost = tok[1], tok[2], 0.0, tok[3], float(tok[4])
                else:
                    raise InvalidProblemError(
                        f"Line {line_no}: Invalid arc descriptor format. Expected 'a <tail> <head> <lower> "
                        f"<capacity> <cost>' or 'a <tail> <head> <capacity> <cost>', got: {line}")
                if cap_s == "-1" or cap_s.lower() == "inf":
                    cap = None
                else:
                    cap_v = float(cap_s)
                    cap = None if cap_v >= 1e15 else cap_v
                arcs.append((tail, head, lower, cap, cost))
            else:
                raise InvalidProblemError(
                    f"Line {line_no}: Unknown line type '{kind}'. Expected 'c' (comment), 'p' (problem), "
                    f"'n' (node), or 'a' (arc).")
        except (ValueError, IndexError) as exc:
            raise InvalidProblemError(f"Line {line_no}: Failed to parse line: {line}. Error: {exc}") from exc
    if not seen_p:
        raise InvalidProblemError(
            "No problem descriptor found. DIMACS file must contain a 'p min <nodes> <arcs>' line.")
    if n_arcs != len(arcs):
        raise InvalidProblemError(
            f"Arc count mismatch: problem descriptor specifies {n_arcs} arcs, but {len(arcs)} arc descriptors found.")
    valid = {str(i) for i in range(1, n_nodes + 1)}
    stray = {t for t, *_ in arcs} | {a[1] for a in arcs}
    stray -= valid
    if stray:
        raise InvalidProblemError(
            f"Arc references node IDs outside the expected range [1, {n_nodes}]: {sorted(stray)}")
    return n_nodes, supplies, arcs


def _parse_lines(lines) -> NetworkProblem:
    n_nodes, supplies, arcs = _scan(lines)
    nodes = [{"id": str(i), "supply": supplies.get(str(i), 0.0)} for i in range(1, n_nodes + 1)]
    arc_dicts = [{"tail": t, "head": h, "lower": lo, "capacity": cap, "cost": c} for t, h, lo, cap, c in arcs]
    return build_problem(nodes=nodes, arcs=arc_dicts, directed=True, tolerance=1e-6)


def parse_dimacs_string(dimacs_content: str) -> NetworkProblem:
    """dimacs.py:46-74."""
    return _parse_lines(dimacs_content.strip().split("\n"))


def parse_dimacs_file(file_path: str | Path) -> NetworkProblem:
    """dimacs.py:77-102."""
    path = Path(file_path)
    if not path.exists():
        raise FileNotFoundError(f"DIMACS file not found: {file_path}")
    with path.open("r", encoding="utf-8") as fh:
        return _parse_lines([ln.rstrip("\n\r") for ln in fh])


def parse_dimacs_soa(file_path: str | Path) -> ArcSoA:
    """DIMACS text -> flat integer arrays (0-based ids) through the library's native reader
    (``mcf_dimacs_scan`` / ``mcf_dimacs_load``): no per-arc Python objects, so it scales to the
    16 M-arc instances the object model cannot hold.
    Integer data and zero lower bounds only -- what netgen / gridgen / goto emit; anything else
    should go through ``parse_dimacs_file``."""
    from . import engine

    path = Path(file_path)
    if not path.exists():
        raise FileNotFoundError(f"DIMACS file not found: {file_path}")
    n, tail, head, lower, cap, cost, supply = engine.dimacs_load(str(path))
    if lower.any():
        raise InvalidProblemError("parse_dimacs_soa supports zero lower bounds only; use parse_dimacs_file.")
    return ArcSoA(n, tail, head, cost, cap, supply, name=path.name)
