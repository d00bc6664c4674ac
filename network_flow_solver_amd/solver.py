"""Public entry points with the reference's signatures (/root/reference/src/network_solver/solver.py)."""

from __future__ import annotations

from pathlib import Path

from .data import Basis, FlowResult, NetworkProblem, ProgressCallback, SoAProblem, SolverOptions
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


# solve_many: problems up to this many nodes run as ONE persistent workgroup each when there are enough of them to fill
# CUs (profiles/r02_batched_*: 256 instances of 1 024 / 4 096 / 8 192 nodes -> 11 M / 6.4 M / 4.7 M pivots/s in total,
# against ~52 K one after the other; a single instance of 16 384 nodes runs at ~0.55x its kernel-per-phase rate this way)
_BATCH_MAX_NODES = 16384
_BATCH_MIN_PROBLEMS = 4


def solve_many(problems, options: SolverOptions | None = None, max_iterations: int | None = None,
               return_exceptions: bool = False) -> list:
    """Solve independent problems side by side: every problem that can run as ONE persistent workgroup -- the whole instance
    in a CU's LDS (up to about 300 nodes / 2 500 arcs) or its state in global memory (up to 16 384 nodes) -- gets one workgroup of one batched launch (``mcf_solve_batch``).  A single such solve can
    only occupy one of the chip's 256 CUs; a batch fills it (measured: 256 netgen_8_08a-sized solves in 5.4 ms, 256
    netgen_8_12a-sized ones in 0.40 s).  Everything else is solved one after the other.  Each result is what
    ``solve_min_cost_flow`` returns for that problem; ``UnboundedProblemError`` / ``InvalidProblemError`` are raised unless
    ``return_exceptions`` (then they take the problem's place in the list).  No reference counterpart: it solves its
    benchmark instances in a Python loop (benchmarks/runners/run_benchmark.py)."""
    from . import engine as _engine

    solvers: list = []
    out: list = [None] * len(problems)
    many = len(problems) >= _BATCH_MIN_PROBLEMS
    try:
        for k, problem in enumerate(problems):
            try:
                n_nodes = int(problem.supply.shape[0]) if isinstance(problem, SoAProblem) else len(problem.nodes)
                loop = {"mid_loop": 1} if many and n_nodes <= _BATCH_MAX_NODES else None
                solvers.append(NetworkSimplex(problem, options=options, engine_options=loop))
            except Exception as exc:  # noqa: BLE001  (InvalidProblemError and friends)
                if not return_exceptions:
                    raise
                solvers.append(None)
                out[k] = exc
        batched = {k for k, sv in enumerate(solvers) if sv is not None and sv.engine.stats()["pricing_mode"] in (2, 3)}
        if batched:
            order = sorted(batched)
            _engine.solve_batch([solvers[k].engine for k in order], [solvers[k].default_budget(max_iterations) for k in order])
        for k, sv in enumerate(solvers):
            if sv is None:
                continue
            try:
                out[k] = sv._collect() if k in batched else sv.solve(max_iterations=max_iterations)
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
