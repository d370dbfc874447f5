"""Scenario / solver / tracker parameters - host-side mirror of the reference's ``marlpde/parameters.py``.

Same names, defaults and derived values as the reference (``Scenario`` :9-48, ``Map_Scenario`` :50-148,
``Solver`` :201-240, ``Tracker`` :243-261) so that ``asdict(...)`` dictionaries can be passed to
:func:`Evolve_scenario.integrate_equations` unchanged.  Differences, all deliberate:

* no ``pint``: the reference attaches units and strips them again with ``.magnitude`` before any
  arithmetic (:99); here the magnitudes are stored directly and the unit is kept as documentation;
* ``Solver.__post_init__`` does not delete entries from the class-level ``__dataclass_fields__``
  (in the reference only the first instance behaves as intended);
* ``Solver.backend`` additionally accepts ``"hip"`` (the default here).

The values are pinned by tests/golden/params_default.json (dumped from the reference).
"""
from dataclasses import dataclass, field, fields, make_dataclass

import numpy as np


@dataclass
class Scenario:
    """Scenario A of L'Heureux (2018), FORTRAN names (reference parameters.py:16-48). Units in comments."""
    mua: float = 100.09            # g/mol
    rhoa: float = 2.95             # g/cm^3
    rhoc: float = 2.71             # g/cm^3
    rhot: float = 2.8              # g/cm^3
    rhow: float = 1.023            # g/cm^3
    D0ca: float = 131.9            # cm^2/a
    D0co3: float = 272.6           # cm^2/a
    Ka: float = 10 ** (-6.19)      # M^2
    Kc: float = 10 ** (-6.37)      # M^2
    beta: float = 0.1              # cm/a
    b: float = 5.0                 # 1/kPa
    k1: float = 1.0                # 1/a
    k2: float = 1.0                # 1/a
    k3: float = 0.1                # 1/a
    k4: float = 0.1                # 1/a
    nn: float = 2.8
    m: float = 2.48
    S: float = 0.1                 # cm/a
    phiinf: float = 0.01
    phi0: float = 0.8
    ca0: float = 0.326e-3          # M
    co30: float = 0.326e-3         # M
    ccal0: float = 0.3
    cara0: float = 0.6
    xdis: float = 50.0             # cm, start of the dissolution zone
    length: float = 500.0          # cm
    Th: float = 100.0              # cm, height of the dissolution zone
    phi00: float = 0.8
    ca00: float = 0.326e-3         # M
    co300: float = 0.326e-3        # M
    ccal00: float = 0.3
    cara00: float = 0.6


# FORTRAN name -> Matlab/Python name (reference parameters.py:60-92), in the Scenario's field order
_MAPPING = {
    "Ka": "KA", "Kc": "KC", "cara0": "CA0", "cara00": "CAIni", "ccal0": "CC0", "ccal00": "CCIni",
    "ca0": "ca0", "ca00": "ca00", "co30": "co30", "co300": "co300", "phi0": "Phi0", "phi00": "PhiIni",
    "xdis": "ShallowLimit", "Th": "Th", "S": "sedimentationrate", "m": "m1", "nn": "n1",
    "rhoa": "rhoa", "rhoc": "rhoc", "rhot": "rhot", "rhow": "rhow", "beta": "beta", "b": "b",
    "D0ca": "D0Ca", "k1": "k1", "k2": "k2", "k3": "k3", "k4": "k4", "mua": "muA", "D0co3": "DCO3",
    "phiinf": "PhiInfty", "length": "max_depth",
}

_DERIVED = ("cCa0", "cCaIni", "cCO30", "cCO3Ini", "DeepLimit", "rhos0", "rhos", "Xstar", "Tstar",
            "m2", "n2", "DCa", "PhiNR", "N", "FV_switch")


def _finish(self):
    """The conversions of the reference's ``post_init`` (parameters.py:120-143)."""
    root_kc = np.sqrt(self.KC)
    self.cCa0 = self.ca0 / root_kc
    self.cCaIni = self.ca00 / root_kc
    self.cCO30 = self.co30 / root_kc
    self.cCO3Ini = self.co300 / root_kc
    self.DeepLimit = self.ShallowLimit + self.Th
    self.rhos0 = self.rhoa * self.CA0 + self.rhoc * self.CC0 + self.rhot * (1 - (self.CA0 + self.CC0))
    self.rhos = self.rhos0
    self.Xstar = self.D0Ca / self.sedimentationrate
    self.Tstar = self.Xstar / self.sedimentationrate
    self.b = self.b / 1e4
    self.m2 = self.m1
    self.n2 = self.n1
    self.DCa = self.D0Ca
    self.PhiNR = self.PhiIni
    self.N = 200          # number of grid cells
    self.FV_switch = 1    # 1: Fiadeiro-Veronis weighting of the solute / porosity gradients


def Map_Scenario():
    """Return the mapped parameter object; ``asdict`` of it is the reference's ``pde_parms`` (47 keys)."""
    defaults = Scenario()
    spec = [(_MAPPING[f.name], float, field(default=getattr(defaults, f.name)))
            for f in fields(Scenario) if f.name in _MAPPING]
    spec += [(name, int if name in ("N", "FV_switch") else float, field(default=None)) for name in _DERIVED]
    cls = make_dataclass("Mapped_parameters", spec, namespace={"__post_init__": _finish})
    return cls()


def jacobian_sparsity(no_depths=None):
    """27-diagonal sparsity pattern for implicit scipy methods (reference parameters.py:150-199).

    Not used by the explicit HIP path; provided so ``Solver(method="Radau")`` keeps its meaning when the
    HIP RHS is driven by scipy."""
    from scipy.sparse import csr_matrix, dia_matrix, lil_matrix
    n_cells = Map_Scenario().N if no_depths is None else no_depths
    n = 5 * n_cells
    offsets = [o + d for o in range(-n + n_cells, n - n_cells + 1, n_cells) for d in (-1, 0, 1)]
    pattern = lil_matrix(dia_matrix((np.ones((len(offsets), n)), offsets), shape=(n, n)))
    pattern[:2 * n_cells, 4 * n_cells:] = 0
    return csr_matrix(pattern)


@dataclass
class Solver:
    """Solver settings (reference parameters.py:201-240)."""
    first_step: float = 1e-6
    atol: float = 1e-3
    rtol: float = 1e-3
    t_span: tuple = (0, 1)          # in units of T*
    method: str = "Radau"           # any scipy.integrate.solve_ivp method; "RK45" runs fully on the GPU
    lband: int = 1                  # LSODA only
    uband: int = 1                  # LSODA only
    backend: str = "hip"            # "hip": RHS / time loop on the MI355X (reference: "numba" | "numpy")
    dense_output: bool = False
    jac_sparsity: object = None

    def __post_init__(self):
        if self.method in ("Radau", "BDF") and self.jac_sparsity is None:
            self.jac_sparsity = jacobian_sparsity()

    def solve_ivp_options(self):
        """The keyword arguments the reference forwards to solve_ivp for this method (parameters.py:228-238)."""
        drop = {"backend"}
        drop |= {"jac_sparsity"} if self.method == "LSODA" else {"lband", "uband"}
        if self.method not in ("Radau", "BDF", "LSODA"):
            drop |= {"jac_sparsity"}
        return {f.name: getattr(self, f.name) for f in fields(self) if f.name not in drop}


@dataclass
class Tracker:
    """Progress / storage settings (reference parameters.py:243-261)."""
    no_progress_updates: int = 100_000
    no_t_eval: int = 2              # 2: only initial and final state
    t_eval: np.ndarray = None

    def __post_init__(self):
        if self.t_eval is None:
            self.t_eval = np.linspace(*Solver.__dataclass_fields__["t_span"].default, num=self.no_t_eval)
