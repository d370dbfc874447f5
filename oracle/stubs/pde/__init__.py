"""Minimal numpy restatement of the py-pde 0.32 pieces the reference touches.

TEST INFRASTRUCTURE - used ONLY by oracle/make_goldens.py in the build container, never
imported by the product path.

py-pde (pinned 0.32.2 in the reference's poetry.lock:955-956) is neither vendored in the
reference nor installed here, so its *published* semantics are restated:

* ``CartesianGrid([[0, L]], [N], periodic=False)``: cell-centred, dx = L/N, centres (i+1/2)dx.
* operators act on an N+2 array whose two end entries are virtual (ghost) cells set from the
  boundary conditions:  {"value": v} -> 2v - u_adj ; {"derivative": d} -> u_adj + dx*d ;
  {"curvature": c} -> c*dx^2 + 2*u_adj - u_adj2.
* forward / backward one-sided differences and the 3-point Laplacian.

The restatement is pinned numerically: the reference's driver run through these stubs
reproduces the three HDF5 goldens written with the real py-pde (tests/test_oracle_goldens.py).
At the last-bit level the py-pde layer is "parity unpinned" (its source is unavailable).

Call sites in the reference: marlpde/LHeureux_model.py:1-2,19-30,96-111,174-178,202-267,288;
marlpde/Evolve_scenario.py:11-12,40-54,76-86.
"""
import numpy as np

from .grids.operators.cartesian import _make_derivative, _make_laplace

__version__ = "0.32.2-stub"


def _ghosts(bc, u, dx):
    """Return (lower ghost, upper ghost) for a 1-D field ``u`` under the 2-entry BC list."""
    out = []
    for side, spec in enumerate(bc):
        adj, adj2 = (u[0], u[1]) if side == 0 else (u[-1], u[-2])
        (kind, val), = spec.items()
        if kind == "value":
            out.append(2.0 * val - adj)
        elif kind == "derivative":
            out.append(adj + dx * val)
        elif kind == "curvature":
            out.append(val * dx**2 + 2.0 * adj - adj2)
        else:
            raise ValueError(f"unsupported boundary condition {spec!r}")
    return out


class CartesianGrid:
    def __init__(self, bounds, shape, periodic=False):
        assert not periodic and len(bounds) == 1 and len(shape) == 1
        lo, hi = bounds[0]
        n = int(shape[0])
        self.shape = (n,)
        self.axes_bounds = ((float(lo), float(hi)),)
        self.discretization = np.array([(hi - lo) / n])
        centres = lo + (np.arange(n) + 0.5) * self.discretization[0]
        self._axes_coords = (centres,)
        self.axes_coords = self._axes_coords
        self._operators = {"laplace": _make_laplace}

    def register_operator(self, name, factory):
        self._operators[name] = factory

    def make_operator(self, name, bc):
        stencil = self._operators[name](self)
        dx = self.discretization[0]
        n = self.shape[0]

        def apply(arr):
            padded = np.empty(n + 2)
            padded[1:-1] = arr
            padded[0], padded[-1] = _ghosts(bc, arr, dx)
            return stencil(padded)

        return apply


def _unwrap(x):
    return x.data if isinstance(x, ScalarField) else x


class ScalarField:
    __array_priority__ = 1000

    def __init__(self, grid, data=0.0, label=None):
        self.grid = grid
        arr = np.asarray(_unwrap(data), dtype=float)
        self.data = np.full(grid.shape, arr) if arr.ndim == 0 else arr.copy()
        self.label = label

    @classmethod
    def from_expression(cls, grid, expression):
        x = grid._axes_coords[0]
        env = {"x": x, "heaviside": np.heaviside, "np": np}
        return cls(grid, eval(expression, {"__builtins__": {}}, env) + 0.0 * x)

    # ufuncs (np.exp, np.log, ...) operate on .data and re-wrap
    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        if method != "__call__":
            return NotImplemented
        res = ufunc(*[_unwrap(i) for i in inputs], **kwargs)
        return ScalarField(self.grid, res)

    def _bin(self, other, op, swap=False):
        a, b = self.data, _unwrap(other)
        return ScalarField(self.grid, op(b, a) if swap else op(a, b))

    def __add__(self, o): return self._bin(o, np.add)
    def __radd__(self, o): return self._bin(o, np.add, True)
    def __sub__(self, o): return self._bin(o, np.subtract)
    def __rsub__(self, o): return self._bin(o, np.subtract, True)
    def __mul__(self, o): return self._bin(o, np.multiply)
    def __rmul__(self, o): return self._bin(o, np.multiply, True)
    def __truediv__(self, o): return self._bin(o, np.true_divide)
    def __rtruediv__(self, o): return self._bin(o, np.true_divide, True)
    def __pow__(self, o): return self._bin(o, np.power)
    def __neg__(self): return ScalarField(self.grid, -self.data)

    def to_scalar(self, func):
        return ScalarField(self.grid, func(self.data))

    def apply_operator(self, name, bc):
        return ScalarField(self.grid, self.grid.make_operator(name, bc)(self.data))

    def laplace(self, bc):
        return self.apply_operator("laplace", bc)


class FieldCollection:
    def __init__(self, fields):
        self.fields = list(fields)
        self.data = np.stack([f.data for f in self.fields])

    def __getitem__(self, i):
        return self.fields[i]
