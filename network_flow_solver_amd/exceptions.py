"""Exception hierarchy with the reference's names and attributes.

Mirrors /root/reference/src/network_solver/exceptions.py (class names, constructor
arguments and public attributes), so ``except`` clauses written against the
reference keep working.
"""

from __future__ import annotations


class NetworkSolverError(Exception):
    """Base class of every error this package raises (exceptions.py:6-17)."""


class InvalidProblemError(NetworkSolverError):
    """Malformed problem definition (exceptions.py:20-36)."""


class InfeasibleProblemError(NetworkSolverError):
    """No feasible flow exists (exceptions.py:39-62). The solver itself reports
    infeasibility through ``FlowResult.status``; this class exists for callers."""

    def __init__(self, message: str, iterations: int = 0):
        super().__init__(message)
        self.iterations = iterations


class UnboundedProblemError(NetworkSolverError):
    """Negative-cost cycle with unlimited capacity (exceptions.py:65-93)."""

    def __init__(self, message: str, entering_arc: tuple[str, str] | None = None,
                 reduced_cost: float | None = None):
        super().__init__(message)
        self.entering_arc = entering_arc
        self.reduced_cost = reduced_cost


class NumericalInstabilityError(NetworkSolverError):
    """Kept for API compatibility (exceptions.py:96-121); the integer engine never raises it."""

    def __init__(self, message: str, condition_number: float | None = None):
        super().__init__(message)
        self.condition_number = condition_number


class IterationLimitError(NetworkSolverError):
    """Optional strict-mode error (exceptions.py:124-154)."""

    def __init__(self, message: str, iterations: int = 0, objective: float | None = None,
                 status: str = "unknown"):
        super().__init__(message)
        self.iterations = iterations
        self.objective = objective
        self.status = status


class SolverConfigurationError(NetworkSolverError):
    """Invalid solver configuration (exceptions.py:157-167)."""
