"""Benchmark-framework adapter: the MI355X engine as a ``SolverAdapter``.

The reference's comparison harness (``/root/reference/benchmarks/solvers/base.py:9-79``) wants a
class with ``name`` / ``display_name`` / ``description`` and the classmethods ``solve(problem,
timeout_s) -> SolverResult``, ``is_available()`` and ``get_version()``; adapters never raise --
failures become ``status="error"`` (``base.py:52-54``).  ``SolverResult`` is restated here with
the same fields so the module works without the reference tree on the path; registering it in the
reference is one line in ``benchmarks/solvers/__init__.py:27``.

The problem object may be this package's ``NetworkProblem`` or the reference's: only the
``directed / nodes / arcs / tolerance`` attributes are read.
"""

from __future__ import annotations

import time
from dataclasses import dataclass

from . import __version__, engine
from .data import Arc, NetworkProblem, Node, SolverOptions
from .exceptions import UnboundedProblemError
from .solver import solve_many, solve_min_cost_flow


@dataclass
class SolverResult:
    """Same fields as benchmarks/solvers/base.py:9-22."""

    solver_name: str
    problem_name: str
    status: str  # 'optimal', 'infeasible', 'timeout', 'error', ...
    objective: float | None
    solve_time_ms: float
    iterations: int | None
    error_message: str | None = None
    metadata: dict | None = None


def _as_native(problem) -> NetworkProblem:
    if isinstance(problem, NetworkProblem):
        return problem
    nodes = {str(k): Node(id=str(v.id), supply=float(v.supply)) for k, v in problem.nodes.items()}
    arcs = [Arc(tail=str(a.tail), head=str(a.head), capacity=a.capacity, cost=float(a.cost), lower=float(a.lower))
            for a in problem.arcs]
    return NetworkProblem(directed=bool(problem.directed), nodes=nodes, arcs=arcs, tolerance=float(problem.tolerance))


class Mi355xAdapter:
    """Adapter for the MI355X network-simplex engine (duck-typed ``SolverAdapter``)."""

    name = "network_solver_mi355x"
    display_name = "Network Solver (MI355X)"
    description = "Network simplex pivoting on an AMD MI355X: HIP pricing / ratio-test / tree kernels"

    @classmethod
    def solve(cls, problem, timeout_s: float = 60.0) -> SolverResult:
        try:
            native = _as_native(problem)
            deadline = time.perf_counter() + float(timeout_s)
            timed_out = []

            def watch(_info):  # progress callback doubles as the timeout check
                if time.perf_counter() > deadline:
                    timed_out.append(True)
                    raise TimeoutError

            start = time.perf_counter()
            try:
                result = solve_min_cost_flow(native, SolverOptions(), max_iterations=None, progress_callback=watch,
                                             progress_interval=4096)
            except TimeoutError:
                return SolverResult(cls.name, "", "timeout", None, (time.perf_counter() - start) * 1e3, None)
            except UnboundedProblemError as exc:
                return SolverResult(cls.name, "", "unbounded", None, (time.perf_counter() - start) * 1e3, None,
                                    error_message=str(exc))
            elapsed_ms = (time.perf_counter() - start) * 1e3
            return SolverResult(cls.name, "", result.status, result.objective if result.status == "optimal" else None,
                                elapsed_ms, result.iterations,
                                metadata={"pricing_strategy": f"{SolverOptions().pricing_strategy} (auto: Dantzig on grid-on-torus)",
                                          "has_duals": True, "device": "MI355X (gfx950)"})
        except Exception as exc:  # adapters never raise (base.py:52-54)
            return SolverResult(cls.name, "", "error", None, 0.0, None, error_message=f"{type(exc).__name__}: {exc}")

    @classmethod
    def solve_many(cls, problems, timeout_s: float = 60.0) -> list:
        """A whole benchmark group at once (no counterpart in base.py: the reference's runner loops over ``solve``): the
        problems that fit one workgroup each share one batched launch (``solver.solve_many``).  One ``SolverResult`` per
        problem, in order; ``solve_time_ms`` is the group's wall time divided by the group's size; never raises."""
        try:
            natives = [_as_native(p) for p in problems]
            start = time.perf_counter()
            results = solve_many(natives, SolverOptions(), return_exceptions=True)
            share_ms = (time.perf_counter() - start) * 1e3 / max(len(natives), 1)
            out = []
            for r in results:
                if isinstance(r, UnboundedProblemError):
                    out.append(SolverResult(cls.name, "", "unbounded", None, share_ms, None, error_message=str(r)))
                elif isinstance(r, BaseException):
                    out.append(SolverResult(cls.name, "", "error", None, share_ms, None, error_message=f"{type(r).__name__}: {r}"))
                else:
                    out.append(SolverResult(cls.name, "", r.status, r.objective if r.status == "optimal" else None, share_ms,
                                            r.iterations, metadata={"batched": True, "has_duals": True, "device": "MI355X (gfx950)"}))
            return out
        except Exception as exc:  # adapters never raise (base.py:52-54)
            return [SolverResult(cls.name, "", "error", None, 0.0, None, error_message=f"{type(exc).__name__}: {exc}") for _ in problems]

    @classmethod
    def is_available(cls) -> bool:
        try:
            return engine.device_count() > 0
        except Exception:
            return False

    @classmethod
    def get_version(cls) -> str | None:
        return __version__
