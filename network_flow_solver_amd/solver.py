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


def load_problem(path: str | Path) -> NetworkProblem:
    """Load a problem from the reference's JSON format (solver.py:107-137)."""
    return _load_problem_file(path)


def save_result(path: str | Path, result: FlowResult) -> None:
    """Save a FlowResult as JSON (solver.py:140-167)."""
    _save_result_file(path, result)
