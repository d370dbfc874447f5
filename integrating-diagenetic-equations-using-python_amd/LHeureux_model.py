"""The five-field L'Heureux (2018) model on the MI355X - host-side mirror of ``marlpde/LHeureux_model.py``.

``LMAHeureuxPorosityDiff`` keeps the reference class's constructor keywords (LHeureux_model.py:12-16),
its RHS callables ``fun`` / ``fun_numba`` with the ``(t, y, progress_proxy, progress_dt, t0)`` signature
(:162, :290) and its seven event functions (:524-593), but every one of them runs a HIP kernel through
the C ABI (include/marl_hip.h).  The py-pde grid object the reference passes as ``Depths`` is replaced
by :class:`DepthGrid` (cell-centred 1-D grid; the only things the reference reads from it are the cell
count, the spacing and the cell centres, :23-24, :322).

A model may hold a whole batch of parameter sets (a sweep): pass a list of keyword dicts to
:meth:`LMAHeureuxPorosityDiff.batch`.

There is no CPU implementation in this package; without libmarl_hip.so and a GPU construction fails.
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _abi
from ._abi import LAYOUT_FIELD_MAJOR, LAYOUT_TILED, MarlParams, MarlStats, NEVENTS

FIELD_NAMES = ("CA", "CC", "cCa", "cCO3", "Phi")  # order of the state vector (Evolve_scenario.py:76-86)
LABELS = ("ARA", "CAL", "Ca", "CO3", "Po")        # labels of get_state (LHeureux_model.py:138-142)


@dataclass(frozen=True)
class DepthGrid:
    """Cell-centred grid on [0, length] with N cells (stands in for pde.CartesianGrid, Evolve_scenario.py:40)."""
    length: float
    N: int

    @property
    def shape(self):
        return (self.N,)

    @property
    def discretization(self):
        return np.array([self.length / self.N])

    @property
    def axes_coords(self):
        return ((np.arange(self.N) + 0.5) * (self.length / self.N),)

    _axes_coords = axes_coords


def _as_ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class RK45Result:
    """What the reference reads off scipy's OdeResult (Evolve_scenario.py:112-178), for one instance."""

    def __init__(self, stats, t=None, y=None, t_events=None):
        self.nfev = int(stats.nfev)
        self.njev = int(stats.njev)
        self.nlu = int(stats.nlu)
        self.n_accepted = int(stats.n_accepted)
        self.n_rejected = int(stats.n_rejected)
        # library status -> scipy status: 0 finished; -1 step size too small; 2 attempt budget (no scipy analogue)
        self.status = int(stats.status)
        self.success = self.status == 0
        self.message = {0: "The solver successfully reached the end of the integration interval.",
                        -1: "Required step size is less than spacing between numbers.",
                        2: "Attempt budget exhausted before the end of the interval."}.get(self.status, "?")
        self.t_reached = float(stats.t)
        self.h_next = float(stats.h_next)
        self.event_values = np.array(stats.event_value[:])
        self.n_events = np.array(stats.n_events[:])
        self.t = t
        self.y = y
        self.t_events = t_events


def pack_blocks(instances, length):
    """The marl_params blocks (include/marl_params.h) of a list of instances (constructor-keyword dicts) on a column of `length`:
    what marl_ctx_create / marl_ctx_set_params receive.  A pure function of its arguments - a rank's blocks are the corresponding
    blocks of the unsharded sweep (tests/test_multi_rank_cpu.py)."""
    blocks = (MarlParams * len(instances))()
    for blk, inst in zip(blocks, instances):
        for name in _abi.PARAM_DOUBLES[:30]:
            setattr(blk, name, float(inst[name]))
        blk.length = float(length)
        blk.shallow_limit = float(inst["ShallowLimit"]) / float(inst["Xstar"])
        blk.deep_limit = float(inst["DeepLimit"]) / float(inst["Xstar"])
        blk.FV_switch = int(inst["FV_switch"])
        blk.dPhi_variable = int(bool(inst.get("dPhi_variable", False)))
    return blocks


def instance_kwargs(pde_parms, instances):
    """Constructor keywords of every instance of a sweep: the scenario dict overridden by each instance's entries, reduced to what the
    constructor takes (what :meth:`LMAHeureuxPorosityDiff.from_scenario` passes on)."""
    import inspect
    names = {p for p in inspect.signature(LMAHeureuxPorosityDiff.__init__).parameters} - {"self", "Depths", "device"}
    sig = inspect.signature(LMAHeureuxPorosityDiff.__init__).parameters
    defaults = {k: sig[k].default for k in names if sig[k].default is not inspect.Parameter.empty and not k.startswith("_")
                and k not in ("slices_all_fields", "not_too_shallow", "not_too_deep")}
    return [defaults | {k: v for k, v in (pde_parms | inst).items() if k in names} for inst in instances]


class LMAHeureuxPorosityDiff:
    """Model object: parameters on the device + the callables scipy / the integrators need."""

    no_fields = 5

    def __init__(self, Depths, slices_all_fields=None, not_too_shallow=None, not_too_deep=None, *,
                 CA0, CC0, cCa0, cCO30, Phi0, sedimentationrate, Xstar, Tstar, k1, k2, k3, k4, m1, m2, n1, n2,
                 b, beta, rhos, rhow, rhos0, KA, KC, muA, D0Ca, PhiNR, PhiInfty, PhiIni, DCa, DCO3, FV_switch,
                 ShallowLimit=50.0, DeepLimit=150.0, dPhi_variable=False, device=0, _extra_instances=()):
        """Same keywords as the reference constructor.  ``not_too_shallow`` / ``not_too_deep`` (py-pde mask
        fields in the reference) are replaced by the two numbers that define them, ``ShallowLimit`` and
        ``DeepLimit`` in cm (Evolve_scenario.py:51-54); ``slices_all_fields`` is implied by N.
        ``dPhi_variable=True`` switches from the reference's fixed porosity diffusion coefficient (:124-133, :431) to
        the time-varying one, auxcon F Phi^3/(1-Phi), that the reference keeps commented out (:222-223, :430)."""
        if not isinstance(Depths, DepthGrid):
            raise TypeError("Depths must be a DepthGrid(length=max_depth/Xstar, N=number of cells)")
        if not_too_shallow is not None or not_too_deep is not None:
            raise TypeError("pass ShallowLimit / DeepLimit (cm) instead of mask fields")
        self.Depths = Depths
        N = Depths.N
        self.slices_all_fields = slices_all_fields or [slice(i * N, (i + 1) * N) for i in range(5)]
        (self.CA_sl, self.CC_sl, self.cCa_sl, self.cCO3_sl, self.Phi_sl) = self.slices_all_fields
        first = dict(CA0=CA0, CC0=CC0, cCa0=cCa0, cCO30=cCO30, Phi0=Phi0, sedimentationrate=sedimentationrate,
                     Xstar=Xstar, Tstar=Tstar, k1=k1, k2=k2, k3=k3, k4=k4, m1=m1, m2=m2, n1=n1, n2=n2, b=b,
                     beta=beta, rhos=rhos, rhow=rhow, rhos0=rhos0, KA=KA, KC=KC, muA=muA, D0Ca=D0Ca,
                     PhiNR=PhiNR, PhiInfty=PhiInfty, PhiIni=PhiIni, DCa=DCa, DCO3=DCO3, FV_switch=FV_switch,
                     ShallowLimit=ShallowLimit, DeepLimit=DeepLimit, dPhi_variable=bool(dPhi_variable))
        self.instances = [first, *_extra_instances]
        for k, v in first.items():   # the reference exposes its parameters as attributes
            setattr(self, k, v)
        self.last_t = 0.0            # progress-bar helper of the reference (:90)
        self.device = int(device)
        self._lib = _abi.load()
        self._ctx = C.c_void_p()
        blocks = self._pack_blocks()
        rc = self._lib.marl_ctx_create(blocks, len(self.instances), N, self.device, C.byref(self._ctx))
        if rc != 0:
            self._ctx = C.c_void_p()
            _abi.check(None, rc, "marl_ctx_create")
        for name, value in _abi.lab_options():   # (kernel-lab A/B runs only: MARL_HIP_OPTIONS="name=value,..."; bench.py records it)
            self.set_option(name, value)
        # derived constants, named as in the reference (:36-72, :130-133)
        names = ("delta_x", "nu1", "nu2", "KRat", "dCa", "dCO3", "delta", "Da", "lambda_", "auxcon", "rhorat0",
                 "rhorat", "presum", "F_fixed", "dPhi_fixed", "Peclet_min", "Peclet_max", "mask_lo", "mask_hi")
        vals = (C.c_double * 19)()
        self._check(self._lib.marl_get_constants(self._ctx, 0, vals), "marl_get_constants")
        for n_, v in zip(names, vals):
            setattr(self, n_, int(v) if n_.startswith("mask_") else float(v))
        x = Depths.axes_coords[0]
        self.not_too_shallow = np.heaviside(x - ShallowLimit / Xstar, 0)
        self.not_too_deep = np.heaviside(DeepLimit / Xstar - x, 0)

    def _pack_blocks(self):
        return pack_blocks(self.instances, self.Depths.length)

    def set_scenario(self, pde_parms):
        """The parameters of another scenario (``asdict(Map_Scenario())``-style dict, same N and max_depth / Xstar) for this
        single-instance model: what constructing the reference's model again does, with the device buffers kept (marl_ctx_set_params)."""
        import inspect
        if len(self.instances) != 1:
            raise ValueError("set_scenario: single-instance models only")
        names = {p for p in inspect.signature(type(self).__init__).parameters} - {"self", "Depths", "device", "slices_all_fields", "not_too_shallow",
                                                                                   "not_too_deep", "_extra_instances"}
        new = {k: v for k, v in pde_parms.items() if k in names}
        if int(pde_parms.get("N", self.Depths.N)) != self.Depths.N:
            raise ValueError("set_scenario: the grid size cannot change")
        # the column's length max_depth / Xstar (Evolve_scenario.py:27-31) may change with the scenario: the grid - and with it delta_x,
        # the masks and the cell centres - is rebuilt, exactly as constructing the model again would (marl_ctx_set_params re-derives
        # every constant from the new length)
        if "max_depth" in pde_parms and "Xstar" in pde_parms:
            length = float(pde_parms["max_depth"]) / float(pde_parms["Xstar"])
            if length != self.Depths.length:
                self.Depths = DepthGrid(length, self.Depths.N)
        sig = inspect.signature(type(self).__init__).parameters
        defaults = {k: sig[k].default for k in names if sig[k].default is not inspect.Parameter.empty}   # as a fresh construction would take them
        missing = [k for k in names if k not in new and k not in defaults]
        if missing:
            raise TypeError(f"set_scenario: missing parameters {sorted(missing)}")
        inst = defaults | new
        inst["dPhi_variable"] = bool(inst.get("dPhi_variable", False))
        self.instances = [inst]
        for k, v in inst.items():
            setattr(self, k, v)
        self._check(self._lib.marl_ctx_set_params(self._ctx, self._pack_blocks(), 1), "marl_ctx_set_params")
        names19 = ("delta_x", "nu1", "nu2", "KRat", "dCa", "dCO3", "delta", "Da", "lambda_", "auxcon", "rhorat0",
                   "rhorat", "presum", "F_fixed", "dPhi_fixed", "Peclet_min", "Peclet_max", "mask_lo", "mask_hi")
        vals = (C.c_double * 19)()
        self._check(self._lib.marl_get_constants(self._ctx, 0, vals), "marl_get_constants")
        for n_, v in zip(names19, vals):
            setattr(self, n_, int(v) if n_.startswith("mask_") else float(v))
        x = self.Depths.axes_coords[0]
        self.not_too_shallow = np.heaviside(x - inst["ShallowLimit"] / inst["Xstar"], 0)
        self.not_too_deep = np.heaviside(inst["DeepLimit"] / inst["Xstar"] - x, 0)

    # -- construction helpers ------------------------------------------------------------------
    @classmethod
    def batch(cls, Depths, instances, device=0):
        """A sweep: ``instances`` is a list of keyword dicts (each as for the constructor)."""
        instances = list(instances)
        return cls(Depths, **instances[0], device=device, _extra_instances=tuple(instances[1:]))

    @classmethod
    def from_scenario(cls, pde_parms, device=0, instances=None):
        """Build from ``asdict(Map_Scenario())``-style dict(s) the way Evolve_scenario.py:27-68 does."""
        import inspect
        names = {p for p in inspect.signature(cls.__init__).parameters} - {"self", "Depths", "device"}
        grid = DepthGrid(pde_parms["max_depth"] / pde_parms["Xstar"], int(pde_parms["N"]))
        pick = lambda d: {k: v for k, v in d.items() if k in names}  # noqa: E731
        if instances is None:
            return cls(grid, **pick(pde_parms), device=device)
        return cls.batch(grid, [pick(pde_parms | inst) for inst in instances], device=device)

    @property
    def n_instances(self):
        return len(self.instances)

    def _check(self, rc, what):
        return _abi.check(self._ctx, rc, what)

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.marl_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, name, value):
        self._check(self._lib.marl_set_option(self._ctx, name.encode(), int(value)), "marl_set_option")

    def use_stream(self, stream_handle):
        """Run on an existing HIP stream, e.g. ``torch.cuda.current_stream().cuda_stream``."""
        self._check(self._lib.marl_set_stream(self._ctx, C.c_void_p(stream_handle)), "marl_set_stream")

    def synchronize(self):
        self._check(self._lib.marl_synchronize(self._ctx), "marl_synchronize")

    def state_doubles(self, layout=LAYOUT_FIELD_MAJOR):
        return int(self._lib.marl_state_doubles(self._ctx, layout))

    def get_state(self, AragoniteSurface, CalciteSurface, CaSurface, CO3Surface, PorSurface):
        """Initial state (5, N) from five uniform values or arrays (reference :135-145)."""
        N = self.Depths.N
        return np.stack([np.broadcast_to(np.asarray(v, dtype=float), (N,)) for v in
                         (AragoniteSurface, CalciteSurface, CaSurface, CO3Surface, PorSurface)]).copy()

    # -- RHS: the solve_ivp callable -------------------------------------------------------------
    def _host_state(self, y):
        y = np.ascontiguousarray(y, dtype=np.float64)
        want = 5 * self.Depths.N * self.n_instances
        if y.size != want:
            raise ValueError(f"state has {y.size} entries, expected {want} (5 fields x N x instances)")
        return y

    def fun(self, t, y, progress_proxy=None, progress_dt=None, t0=None):
        """dy/dt for the field-major state ``y`` (host array); returns a new array (reference :162-288)."""
        if progress_proxy is not None and progress_dt:   # the reference's tqdm hook (:168-172)
            if self.last_t == 0.0:
                self.last_t = t0
            n = int((t - self.last_t) / progress_dt)
            progress_proxy.update(n)
            self.last_t += n * progress_dt
        y = self._host_state(y)
        out = np.empty_like(y)
        self._check(self._lib.marl_rhs(self._ctx, float(t), _as_ptr(y), _as_ptr(out)), "marl_rhs")
        return out

    fun_numba = fun  # the reference's second spelling of the same callable (:290-359)

    def rhs_device(self, y_dev_ptr, dydt_dev_ptr, layout=LAYOUT_FIELD_MAJOR, t=0.0):
        """RHS on device buffers (raw pointers, e.g. ``tensor.data_ptr()``); asynchronous."""
        self._check(self._lib.marl_rhs_dev(self._ctx, float(t), C.c_void_p(y_dev_ptr), C.c_void_p(dydt_dev_ptr), layout),
                    "marl_rhs_dev")

    # -- monitors (solve_ivp events) -----------------------------------------------------------------
    def events_all(self, y):
        """The seven monitors, shape (instances, 7), in the reference's order (Evolve_scenario.py:107-109)."""
        y = self._host_state(y)
        out = np.empty((self.n_instances, NEVENTS))
        self._check(self._lib.marl_events(self._ctx, _as_ptr(y), _as_ptr(out)), "marl_events")
        return out

    def events_device(self, y_dev_ptr, layout=LAYOUT_FIELD_MAJOR):
        out = np.empty((self.n_instances, NEVENTS))
        self._check(self._lib.marl_events_dev(self._ctx, C.c_void_p(y_dev_ptr), layout, _as_ptr(out)), "marl_events_dev")
        return out

    def _event(index):  # noqa: N805 - builds the seven scipy-style event callables
        def event(self, t, y, progress_proxy=None, progress_dt=None, t0=None):
            return float(self.events_all(y)[0, index])
        event.terminal = False   # reference :116-122: all seven only record, never stop
        return event

    zeros = _event(0)
    zeros_CA = _event(1)
    zeros_CC = _event(2)
    ones_CA_plus_CC = _event(3)
    ones_Phi = _event(4)
    zeros_U = _event(5)
    zeros_W = _event(6)
    del _event

    def convert_layout_device(self, src_ptr, dst_ptr, src_layout, dst_layout):
        self._check(self._lib.marl_convert_layout_dev(self._ctx, C.c_void_p(src_ptr), C.c_void_p(dst_ptr), src_layout,
                                                      dst_layout), "marl_convert_layout_dev")

    # -- time loops ----------------------------------------------------------------------------------
    def integrate_rk4(self, y, dt, nsteps):
        """Classical RK4, ``nsteps`` steps of ``dt``; host state in, new host state out."""
        y = self._host_state(y).copy()
        self._check(self._lib.marl_integrate_rk4(self._ctx, _as_ptr(y), float(dt), int(nsteps)), "marl_integrate_rk4")
        return y

    def integrate_rk4_device(self, y_dev_ptr, dt, nsteps, layout=LAYOUT_FIELD_MAJOR):
        self._check(self._lib.marl_integrate_rk4_dev(self._ctx, C.c_void_p(y_dev_ptr), layout, float(dt), int(nsteps)),
                    "marl_integrate_rk4_dev")

    def sweep_rk4_device(self, y_dev_ptr, dt, nsteps):
        dt = np.ascontiguousarray(np.broadcast_to(np.asarray(dt, dtype=np.float64), (self.n_instances,)))
        self._check(self._lib.marl_sweep_rk4_dev(self._ctx, C.c_void_p(y_dev_ptr), _as_ptr(dt), int(nsteps)),
                    "marl_sweep_rk4_dev")

    def integrate_rk45(self, y0, t_span, first_step, rtol, atol, t_eval=None, events=True, max_events=4096,
                       max_attempts=0):
        """scipy ``solve_ivp(method="RK45")`` semantics for ONE instance, entirely on the device.

        Returns an :class:`RK45Result` whose ``t``/``y`` hold the ``t_eval`` samples (``y``: (5N, n_t), as
        scipy) or, without ``t_eval``, the end point only; ``t_events`` is a list of 7 arrays."""
        y_start = self._host_state(y0)
        n = y_start.size
        te = None if t_eval is None else np.ascontiguousarray(t_eval, dtype=np.float64)
        n_eval = 0 if te is None else te.size
        while True:
            y = y_start.copy()
            stats = MarlStats()
            y_eval = np.empty((max(n_eval, 1), n))
            tev = np.full((NEVENTS, max_events), np.nan) if events else None
            rc = self._lib.marl_integrate_rk45(
                self._ctx, _as_ptr(y), float(t_span[0]), float(t_span[1]), float(first_step), float(rtol), float(atol),
                _as_ptr(te) if n_eval else None, n_eval, _as_ptr(y_eval) if n_eval else None,
                _as_ptr(tev) if events else None, max_events if events else 0, int(max_attempts), C.byref(stats))
            self._check(rc, "marl_integrate_rk45")
            most = max(stats.n_events[:]) if events else 0
            if most <= max_events:
                break
            # solve_ivp returns EVERY root time (a monitor sitting at exactly 0 fires on every accepted step): the run is
            # deterministic, so repeat it with a buffer that holds them all
            max_events = int(most)
        t_events = None
        if events:
            t_events = [tev[e, :min(int(stats.n_events[e]), max_events)].copy() for e in range(NEVENTS)]
        if n_eval:
            k = int(np.searchsorted(te, stats.t, side="right"))
            res = RK45Result(stats, te[:k].copy(), y_eval[:k].T.copy(), t_events)
        else:
            res = RK45Result(stats, np.array([stats.t]), y[:, None].copy(), t_events)
        res.y_final = y
        return res

    def integrate_radau(self, y0, t_span, first_step, rtol, atol, t_eval=None, events=True, max_events=4096, max_attempts=0,
                        groups=None):
        """scipy ``solve_ivp(method="Radau", jac_sparsity=<the reference's 27-diagonal pattern>)`` semantics for ONE instance -
        the reference's default solver (marlpde/parameters.py:213) - with the RHS, the finite-difference Jacobian, the
        block-tridiagonal factorisations and all vector work on the device (marl_integrate_radau).

        ``groups``: scipy's column grouping of the pattern (``scipy.optimize._numdiff.group_columns``), or None for a
        structured 15-colouring (same Jacobian).  Returns an :class:`RK45Result` (``nfev``/``njev``/``nlu`` as scipy counts)."""
        return self._integrate_implicit("marl_integrate_radau", y0, t_span, first_step, rtol, atol, t_eval, events, max_events, max_attempts, groups)

    def integrate_bdf(self, y0, t_span, first_step, rtol, atol, t_eval=None, events=True, max_events=4096, max_attempts=0, groups=None):
        """scipy ``solve_ivp(method="BDF", jac_sparsity=<the reference's pattern>)`` semantics for ONE instance - the other implicit
        method the reference's Solver names (marlpde/parameters.py:205-219) - on the device (marl_integrate_bdf): variable order 1..5,
        the matrix I - c J factorised by cyclic reduction.  Arguments and result as for :meth:`integrate_radau`."""
        return self._integrate_implicit("marl_integrate_bdf", y0, t_span, first_step, rtol, atol, t_eval, events, max_events, max_attempts, groups)

    def _integrate_implicit(self, entry, y0, t_span, first_step, rtol, atol, t_eval, events, max_events, max_attempts, groups):
        y_start = self._host_state(y0)
        n = y_start.size
        te = None if t_eval is None else np.ascontiguousarray(t_eval, dtype=np.float64)
        n_eval = 0 if te is None else te.size
        grp = None if groups is None else np.ascontiguousarray(groups, dtype=np.int32)
        if grp is not None and grp.size != n:
            raise ValueError(f"groups has {grp.size} entries, expected {n}")
        while True:
            y = y_start.copy()
            stats = MarlStats()
            y_eval = np.empty((max(n_eval, 1), n))
            tev = np.full((NEVENTS, max_events), np.nan) if events else None
            rc = getattr(self._lib, entry)(
                self._ctx, _as_ptr(y), float(t_span[0]), float(t_span[1]), float(first_step), float(rtol), float(atol),
                _as_ptr(grp) if grp is not None else None, _as_ptr(te) if n_eval else None, n_eval, _as_ptr(y_eval) if n_eval else None,
                _as_ptr(tev) if events else None, max_events if events else 0, int(max_attempts), C.byref(stats))
            self._check(rc, entry)
            most = max(stats.n_events[:]) if events else 0
            if most <= max_events:
                break
            max_events = int(most)     # scipy returns every root time: repeat the (deterministic) run with room for all
        t_events = [tev[e, :int(stats.n_events[e])].copy() for e in range(NEVENTS)] if events else None
        if n_eval:
            k = int(np.searchsorted(te, stats.t, side="right"))
            res = RK45Result(stats, te[:k].copy(), y_eval[:k].T.copy(), t_events)
        else:
            res = RK45Result(stats, np.array([stats.t]), y[:, None].copy(), t_events)
        res.y_final = y
        return res

    def integrate_rk45_device(self, y_dev_ptr, t_span, first_step, rtol, atol, layout=LAYOUT_FIELD_MAJOR, max_attempts=0):
        stats = MarlStats()
        rc = self._lib.marl_integrate_rk45_dev(self._ctx, C.c_void_p(y_dev_ptr), layout, float(t_span[0]), float(t_span[1]),
                                               float(first_step), float(rtol), float(atol), int(max_attempts), C.byref(stats))
        self._check(rc, "marl_integrate_rk45_dev")
        return RK45Result(stats)

    def sweep_radau_device(self, y_dev_ptr, t_span, first_step, rtol, atol, max_attempts=0, groups=None, events=False, max_events=64):
        """Every instance of the model integrated with the reference's default solver (scipy Radau semantics), all instances
        advanced together on the device (marl_sweep_radau_dev); device states [instances][5N] in place.  Returns one
        :class:`RK45Result` per instance (statistics).  ``events=True``: the monitors' root times are located inside the sweep
        (marl_sweep_radau_events_dev) and returned as ``t_events`` - a list of 7 arrays per instance, what the reference prints and
        stores for every run (Evolve_scenario.py:118-145, 175-177); otherwise sign changes are only counted."""
        grp = None if groups is None else np.ascontiguousarray(groups, dtype=np.int32)
        stats = (MarlStats * self.n_instances)()
        args = (self._ctx, C.c_void_p(y_dev_ptr), float(t_span[0]), float(t_span[1]), float(first_step), float(rtol), float(atol),
                _as_ptr(grp) if grp is not None else None, int(max_attempts))
        if not events:
            self._check(self._lib.marl_sweep_radau_dev(*args, stats), "marl_sweep_radau_dev")
            return [RK45Result(s) for s in stats]
        tev = np.full((self.n_instances, NEVENTS, int(max_events)), np.nan)
        self._check(self._lib.marl_sweep_radau_events_dev(*args, _as_ptr(tev), int(max_events), stats), "marl_sweep_radau_events_dev")
        return [RK45Result(s, t_events=[tev[b, e, :min(int(s.n_events[e]), int(max_events))].copy() for e in range(NEVENTS)])
                for b, s in enumerate(stats)]

    def sweep_rk45_device(self, y_dev_ptr, t_span, first_step, rtol, atol, max_attempts=0):
        stats = (MarlStats * self.n_instances)()
        rc = self._lib.marl_sweep_rk45_dev(self._ctx, C.c_void_p(y_dev_ptr), float(t_span[0]), float(t_span[1]),
                                           float(first_step), float(rtol), float(atol), int(max_attempts), stats)
        self._check(rc, "marl_sweep_rk45_dev")
        return [RK45Result(s) for s in stats]


__all__ = ["LMAHeureuxPorosityDiff", "DepthGrid", "RK45Result", "FIELD_NAMES", "LAYOUT_FIELD_MAJOR", "LAYOUT_TILED"]
