"""Public entry points with the reference's signatures (/root/reference/src/network_solver/solver.py)."""

from __future__ import annotations

from pathlib import Path

from .data import Basis, FlowResult, NetworkProblem, ProgressCallback, SolverOptions
from .io import load_problem as _load_problem_file
from .io import save_result as _save_result_file
from .simplex import NetworkSimplex


def solve_min_cost_flow(
    problem: NetworkProblem,
    options: SolverOptions | None = None,
    max_iterations: int | None = None,
    progress_callback: ProgressCallback | None = None,
    progress_interval: int = 100,
    warm_start_basis: Basis | None = None,
) -> FlowResult:
    """Solve a minimum-cost flow problem on the MI355X network-simplex engine.

    Drop-in for ``network_solver.solve_min_cost_flow`` (solver.py:13-104): a fresh solver per
    call, ``max_iterations`` overrides ``options.max_iterations``, unbounded problems raise
    ``UnboundedProblemError``, infeasible ones come back with ``status == "infeasible"``.
    """
    solver = NetworkSimplex(problem, options=options)
    try:
        return solver.solve(max_iterations=max_iterations, progress_callback=progress_callback,
                            progress_interval=progress_interval, warm_start_basis=warm_start_basis)
    finally:
        solver.engine.close()


def solve_many(problems, options: SolverOptions | None = None, max_iterations: int | None = None,
               return_exceptions: bool = False) -> list:
    """Solve independent problems; the small ones (whole instance in one CU's LDS: up to about 300 nodes / 2 500 arcs) side
    by side in ONE launch, one persistent workgroup per problem (``mcf_solve_batch``) -- a single such instance can only
    occupy one of the chip's 256 CUs, a batch fills it (measured: 256 netgen_8_08a-sized solves in 5.4 ms).  Larger
    problems are solved one after the other.  Each result is what ``solve_min_cost_flow`` returns for that problem;
    ``UnboundedProblemError`` / ``InvalidProblemError`` are raised unless ``return_exceptions`` (then they take the
    problem's place in the list).  No reference counterpart: it solves its benchmark instances in a Python loop
    (benchmarks/runners/run_benchmark.py)."""
    from . import engine as _engine

    solvers: list = []
    out: list = [None] * len(problems)
    try:
        for k, problem in enumerate(problems):
            try:
                solvers.append(NetworkSimplex(problem, options=options))
            except Exception as exc:  # noqa: BLE001  (InvalidProblemError and friends)
                if not return_exceptions:
                    raise
                solvers.append(None)
                out[k] = exc
        small = [k for k, sv in enumerate(solvers) if sv is not None and sv.engine.stats()["pricing_mode"] == 2]
        if small:
            _engine.solve_batch([solvers[k].engine for k in small], [solvers[k].default_budget(max_iterations) for k in small])
        for k, sv in enumerate(solvers):
            if sv is None:
                continue
            try:
                out[k] = sv._collect() if k in set(small) else sv.solve(max_iterations=max_iterations)
            except Exception as exc:  # noqa: BLE001  (UnboundedProblemError)
                if not return_exceptions:
                    raise
                out[k] = exc
        return out
    finally:
        for sv in solvers:
            if sv is not None:
                sv.engine.close()


def load_problem(path: str | Path) -> NetworkProblem:
    """Load a problem from the reference's JSON format (solver.py:107-137)."""
    return _load_problem_file(path)


def save_result(path: str | Path, result: FlowResult) -> None:
    """Save a FlowResult as JSON (solver.py:140-167)."""
    _save_result_file(path, result)
