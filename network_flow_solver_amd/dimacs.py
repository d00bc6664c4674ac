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

from .data import NetworkProblem, SoAProblem, build_problem
from .exceptions import InvalidProblemError


# Messages are part of the interface (callers and the reference's tests match on them); the reader below is a small
# table-driven state machine: one handler per record kind, the record's shape checked against a spec before the
# handler sees it.
_MSG = {
    "p_twice": "Multiple problem descriptor lines found. Only one 'p min' line is allowed.",
    "p_shape": "Invalid problem descriptor format. Expected 'p min <nodes> <arcs>', got: {line}",
    "p_kind": "Only 'min' (minimum cost flow) problems supported. Got: {kind}",
    "p_nodes": "Number of nodes must be positive, got {n}",
    "p_arcs": "Number of arcs cannot be negative, got {m}",
    "n_early": "Node descriptor before problem descriptor. The 'p min' line must come first.",
    "n_shape": "Invalid node descriptor format. Expected 'n <node_id> <supply>', got: {line}",
    "a_early": "Arc descriptor before problem descriptor. The 'p min' line must come first.",
    "a_shape": "Invalid arc descriptor format. Expected 'a <tail> <head> <lower> <capacity> <cost>' or "
               "'a <tail> <head> <capacity> <cost>', got: {line}",
    "unknown": "Unknown line type '{kind}'. Expected 'c' (comment), 'p' (problem), 'n' (node), or 'a' (arc).",
}


class _DimacsReader:
    """Collects the records of a 'p min' file.  ``feed(line_no, text)`` per line, then ``finish()``."""

    def __init__(self):
        self.n_nodes: int | None = None
        self.n_arcs: int | None = None
        self.supplies: dict[str, float] = {}
        self.arcs: list[tuple[str, str, float, float | None, float]] = []
        self._handlers = {"p": self._problem, "n": self._node, "a": self._arc}

    @staticmethod
    def _fail(line_no: int, key: str, **kw):
        raise InvalidProblemError(f"Line {line_no}: " + _MSG[key].format(**kw))

    def feed(self, line_no: int, raw: str) -> None:
        line = raw.strip()
        if not line or line[0] == "c":
            return
        fields = line.split()
        handler = self._handlers.get(fields[0])
        if handler is None:
            self._fail(line_no, "unknown", kind=fields[0])
        try:
            handler(line_no, line, fields[1:])
        except (ValueError, IndexError) as exc:
            raise InvalidProblemError(f"Line {line_no}: Failed to parse line: {line}. Error: {exc}") from exc

    def _problem(self, line_no: int, line: str, f: list[str]) -> None:
        if self.n_nodes is not None:
            self._fail(line_no, "p_twice")
        if len(f) != 3:
            self._fail(line_no, "p_shape", line=line)
        if f[0] != "min":
            self._fail(line_no, "p_kind", kind=f[0])
        n, m = int(f[1]), int(f[2])
        self.n_nodes, self.n_arcs = n, m            # (recorded first: a bad count still counts as "seen")
        if n <= 0:
            self._fail(line_no, "p_nodes", n=n)
        if m < 0:
            self._fail(line_no, "p_arcs", m=m)

    def _node(self, line_no: int, line: str, f: list[str]) -> None:
        if self.n_nodes is None:
            self._fail(line_no, "n_early")
        if len(f) != 2:
            self._fail(line_no, "n_shape", line=line)
        self.supplies[f[0]] = float(f[1])

    def _arc(self, line_no: int, line: str, f: list[str]) -> None:
        if self.n_nodes is None:
            self._fail(line_no, "a_early")
        if len(f) not in (4, 5):
            self._fail(line_no, "a_shape", line=line)
        lower = float(f[2]) if len(f) == 5 else 0.0          # 4 fields: no lower bound (dimacs.py:194-207)
        cap_text, cost = f[-2], float(f[-1])
        unlimited = cap_text == "-1" or cap_text.lower() == "inf" or float(cap_text) >= 1e15   # dimacs.py:216-221
        self.arcs.append((f[0], f[1], lower, None if unlimited else float(cap_text), cost))

    def finish(self):
        if self.n_nodes is None:
            raise InvalidProblemError(
                "No problem descriptor found. DIMACS file must contain a 'p min <nodes> <arcs>' line.")
        if self.n_arcs != len(self.arcs):
            raise InvalidProblemError(
                f"Arc count mismatch: problem descriptor specifies {self.n_arcs} arcs, but {len(self.arcs)} arc "
                f"descriptors found.")
        known = {str(i) for i in range(1, self.n_nodes + 1)}
        stray = {end for t, h, *_ in self.arcs for end in (t, h)} - known
        if stray:
            raise InvalidProblemError(
                f"Arc references node IDs outside the expected range [1, {self.n_nodes}]: {sorted(stray)}")
        return self.n_nodes, self.supplies, self.arcs


def _scan(lines):
    reader = _DimacsReader()
    for line_no, raw in enumerate(lines, start=1):
        reader.feed(line_no, raw)
    return reader.finish()


def _parse_lines(lines) -> NetworkProblem:
    n_nodes, supplies, arcs = _scan(lines)
    nodes = [{"id": str(i), "supply": supplies.get(str(i), 0.0)} for i in range(1, n_nodes + 1)]
    arc_dicts = [{"tail": t, "head": h, "lower": lo, "capacity": cap, "cost": c} for t, h, lo, cap, c in arcs]
    return build_problem(nodes=nodes, arcs=arc_dicts, directed=True, tolerance=1e-6)


def parse_dimacs_string(dimacs_content: str) -> NetworkProblem:
    """dimacs.py:46-74."""
    return _parse_lines(dimacs_content.strip().split("\n"))


NATIVE_MIN_ARCS = 100_000   # parse_dimacs_file(native=None): files announcing at least this many arcs take the flat path


def _announced_arcs(path: Path) -> int:
    """Arc count of the 'p min <nodes> <arcs>' line (0 when there is none: the object parser then reports it)."""
    with path.open("r", encoding="utf-8", errors="replace") as fh:
        for raw in fh:
            tok = raw.split()
            if tok and tok[0] == "p":
                try:
                    return int(tok[3])
                except (IndexError, ValueError):
                    return 0
    return 0


def parse_dimacs_file(file_path: str | Path, native: bool | None = None) -> NetworkProblem | SoAProblem:
    """dimacs.py:77-102.  ``native`` selects the library's reader (``mcf_dimacs_scan`` / ``mcf_dimacs_load``), which
    returns an ``SoAProblem`` -- flat integer arrays behind the NetworkProblem surface, no Python object per arc, the
    form ``solve_min_cost_flow`` hands to the engine as is.  ``None``: native from NATIVE_MIN_ARCS announced arcs on
    (integer data; a file with non-integral numbers falls back to the object parser), else the reference's object model."""
    path = Path(file_path)
    if not path.exists():
        raise FileNotFoundError(f"DIMACS file not found: {file_path}")
    if native is None:
        native = _announced_arcs(path) >= NATIVE_MIN_ARCS
        if native:
            try:
                return parse_dimacs_soa(path)
            except InvalidProblemError as exc:
                if "integer" not in str(exc):
                    raise
                native = False
    if native:
        return parse_dimacs_soa(path)
    with path.open("r", encoding="utf-8") as fh:
        return _parse_lines([ln.rstrip("\n\r") for ln in fh])


def parse_dimacs_soa(file_path: str | Path) -> SoAProblem:
    """DIMACS text -> ``SoAProblem`` (0-based flat integer arrays, lower bounds included) through the library's native
    reader: no per-arc Python objects, so it scales to the 16 M-arc instances the object model cannot hold
    (SURVEY.md section 8f item 1).  Integer data only -- what netgen / gridgen / goto emit."""
    from . import engine

    path = Path(file_path)
    if not path.exists():
        raise FileNotFoundError(f"DIMACS file not found: {file_path}")
    n, tail, head, lower, cap, cost, supply = engine.dimacs_load(str(path))
    return SoAProblem(n, tail, head, cost, cap, supply, lower=lower, tolerance=1e-6, name=path.name)
