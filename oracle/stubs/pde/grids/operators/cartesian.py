"""One-sided differences and the 3-point Laplacian on a ghost-padded 1-D array.

TEST INFRASTRUCTURE (see oracle/stubs/pde/__init__.py).  Restates the published behaviour of
py-pde 0.32 ``pde.grids.operators.cartesian._make_derivative`` (registered by the reference at
marlpde/LHeureux_model.py:19-22 and marlpde/Evolve_scenario.py:43-46) and of its built-in 1-D
"laplace" operator.  Each factory takes the grid and returns ``f(padded[N+2]) -> out[N]``.
"""
import numpy as np


def _make_derivative(grid, axis=0, method="central"):
    dx = grid.discretization[axis]
    if method == "forward":
        return lambda p: (p[2:] - p[1:-1]) / dx
    if method == "backward":
        return lambda p: (p[1:-1] - p[:-2]) / dx
    if method == "central":
        return lambda p: (p[2:] - p[:-2]) / (2 * dx)
    raise ValueError(f"unknown derivative method {method!r}")


def _make_laplace(grid):
    scale = grid.discretization[0] ** -2
    return lambda p: (p[:-2] - 2 * p[1:-1] + p[2:]) * scale
