"""network_flow_solver_amd -- MI355X-native network-simplex pivot engine behind the API of
jeffreyhorn/network_flow_solver.

    from network_flow_solver_amd import load_problem, solve_min_cost_flow
    result = solve_min_cost_flow(load_problem("examples/sample_problem.json"))

Exports mirror /root/reference/src/network_solver/__init__.py:54-107 for the solve path
(problem model, options, results, loaders, exceptions).  The pivot loop itself lives in
``csrc/`` (HIP kernels + C ABI, ``include/mcf.h``) and is reached through ``engine``.
"""

from .data import (
    Arc,
    Basis,
    FlowResult,
    NetworkProblem,
    Node,
    ProgressCallback,
    ProgressInfo,
    SoAProblem,
    SolverOptions,
    build_problem,
)
from .dimacs import parse_dimacs_file, parse_dimacs_soa, parse_dimacs_string
from .exceptions import (
    InfeasibleProblemError,
    InvalidProblemError,
    IterationLimitError,
    NetworkSolverError,
    NumericalInstabilityError,
    SolverConfigurationError,
    UnboundedProblemError,
)
from .simplex import NetworkSimplex
from .solver import load_problem, save_result, solve_many, solve_min_cost_flow

__all__ = [
    "Arc", "Basis", "FlowResult", "NetworkProblem", "Node", "ProgressCallback", "ProgressInfo", "SoAProblem", "SolverOptions",
    "build_problem", "parse_dimacs_file", "parse_dimacs_soa", "parse_dimacs_string", "InfeasibleProblemError",
    "InvalidProblemError", "IterationLimitError", "NetworkSolverError", "NumericalInstabilityError",
    "SolverConfigurationError", "UnboundedProblemError", "NetworkSimplex", "load_problem", "save_result",
    "solve_many",
    "solve_min_cost_flow",
]

__version__ = "0.1.0"
