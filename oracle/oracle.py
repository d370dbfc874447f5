"""ctypes wrapper of the CPU oracle (oracle/libmarl_oracle.so, source oracle/marl_oracle.c).

TEST INFRASTRUCTURE: the checker for the HIP path and the "port" CPU baseline.  The product package
never imports this module.  It reuses the product's POD parameter block so both sides are fed the
very same bytes.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from marlpde_amd._abi import PARAM_DOUBLES, MarlParams, MarlStats, NEVENTS

HERE = os.path.dirname(os.path.abspath(__file__))
_libs = {}


def build():
    subprocess.run(["make", "-C", HERE], check=True, capture_output=True)


def lib(omp=False):
    name = "libmarl_oracle_omp.so" if omp else "libmarl_oracle.so"
    if name not in _libs:
        path = os.path.join(HERE, name)
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        P, D, I64, VP = C.POINTER(MarlParams), C.c_double, C.c_int64, C.c_void_p
        L.marl_oracle_derive.argtypes = [P, I64, VP]
        L.marl_oracle_rhs.argtypes = [P, I64, VP, VP]
        L.marl_oracle_events.argtypes = [P, I64, VP, VP]
        L.marl_oracle_rk4.argtypes = [P, I64, VP, D, I64]
        L.marl_oracle_rk4.restype = C.c_int
        L.marl_oracle_rk45.argtypes = [P, I64, VP, D, D, D, D, D, VP, I64, VP, VP, I64, C.POINTER(I64), VP, I64, I64,
                                       C.POINTER(MarlStats)]
        L.marl_oracle_rk45.restype = C.c_int
        L.marl_oracle_radau.argtypes = [P, I64, VP, D, D, D, D, D, VP, VP, I64, VP, VP, I64, C.POINTER(I64), VP, I64, I64,
                                        C.POINTER(MarlStats)]
        L.marl_oracle_radau.restype = C.c_int
        L.marl_oracle_bdf.argtypes = L.marl_oracle_radau.argtypes
        L.marl_oracle_bdf.restype = C.c_int
        L.marl_oracle_rk4_batch.argtypes = [P, I64, I64, VP, VP, I64]
        L.marl_oracle_rk4_batch.restype = C.c_int
        _libs[name] = L
    return _libs[name]


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def params_from_dict(p, length=None):
    """marl_params from an ``asdict(Map_Scenario())``-style dict (same packing as LMAHeureuxPorosityDiff)."""
    blk = MarlParams()
    for name in PARAM_DOUBLES[:30]:
        setattr(blk, name, float(p[name]))
    blk.length = float(p["max_depth"] / p["Xstar"] if length is None else length)
    blk.shallow_limit = float(p["ShallowLimit"]) / float(p["Xstar"])
    blk.deep_limit = float(p["DeepLimit"]) / float(p["Xstar"])
    blk.FV_switch = int(p["FV_switch"])
    blk.dPhi_variable = int(p.get("dPhi_variable", 0))
    return blk


def params_from_model(eq, inst=0):
    d = dict(eq.instances[inst])
    blk = MarlParams()
    for name in PARAM_DOUBLES[:30]:
        setattr(blk, name, float(d[name]))
    blk.length = float(eq.Depths.length)
    blk.shallow_limit = float(d["ShallowLimit"]) / float(d["Xstar"])
    blk.deep_limit = float(d["DeepLimit"]) / float(d["Xstar"])
    blk.FV_switch = int(d["FV_switch"])
    blk.dPhi_variable = int(d.get("dPhi_variable", 0))
    return blk


DERIVED_NAMES = ("delta_x", "nu1", "nu2", "KRat", "dCa", "dCO3", "delta", "Da", "lambda_", "auxcon", "rhorat0",
                 "rhorat", "presum", "F_fixed", "dPhi_fixed", "Peclet_min", "Peclet_max")


def derive(P, N):
    out = np.empty(17)
    lib().marl_oracle_derive(C.byref(P), N, _ptr(out))
    return dict(zip(DERIVED_NAMES, out))


def rhs(P, N, y, omp=False):
    y = np.ascontiguousarray(y, dtype=np.float64)
    out = np.empty_like(y)
    lib(omp).marl_oracle_rhs(C.byref(P), N, _ptr(y), _ptr(out))
    return out


def events(P, N, y):
    y = np.ascontiguousarray(y, dtype=np.float64)
    out = np.empty(NEVENTS)
    lib().marl_oracle_events(C.byref(P), N, _ptr(y), _ptr(out))
    return out


def rk4(P, N, y, dt, nsteps, omp=False):
    y = np.array(y, dtype=np.float64)
    rc = lib(omp).marl_oracle_rk4(C.byref(P), N, _ptr(y), dt, nsteps)
    assert rc == 0
    return y


def rk4_batch(Ps, N, y, dts, nsteps, omp=False):
    """Ps: ctypes array of MarlParams; y: (batch, 5N)."""
    y = np.array(y, dtype=np.float64)
    dts = np.ascontiguousarray(dts, dtype=np.float64)
    rc = lib(omp).marl_oracle_rk4_batch(Ps, len(Ps), N, _ptr(y), _ptr(dts), nsteps)
    assert rc == 0
    return y


def rk45(P, N, y0, t0, t1, first_step, rtol, atol, t_eval=None, max_steps_out=1 << 20, max_events=256,
         max_attempts=0, omp=False):
    """Returns (y_final, stats, step_times, y_eval, t_events)."""
    y = np.array(y0, dtype=np.float64)
    st = MarlStats()
    te = None if t_eval is None else np.ascontiguousarray(t_eval, dtype=np.float64)
    n_eval = 0 if te is None else te.size
    y_eval = np.empty((max(n_eval, 1), y.size))
    steps = np.empty(max_steps_out)
    nsteps = C.c_int64(0)
    tev = np.full((NEVENTS, max_events), np.nan)
    lib(omp).marl_oracle_rk45(C.byref(P), N, _ptr(y), t0, t1, first_step, rtol, atol,
                              _ptr(te) if n_eval else None, n_eval, _ptr(y_eval), _ptr(steps), max_steps_out,
                              C.byref(nsteps), _ptr(tev), max_events, max_attempts, C.byref(st))
    t_events = [tev[e, :min(int(st.n_events[e]), max_events)].copy() for e in range(NEVENTS)]
    return y, st, steps[:min(nsteps.value, max_steps_out)].copy(), y_eval[:n_eval], t_events


def scipy_groups(N):
    """The column grouping scipy derives for the reference's 27-diagonal Jacobian pattern (marlpde/parameters.py:150-199:
    field-major, CA/CC rows x Phi columns zeroed) - scipy.optimize._numdiff.group_columns, as Radau._validate_jac calls it
    (radau.py:349-353).  scipy is a third-party dependency of the reference, importable here and on the GPU box."""
    from scipy.optimize._numdiff import group_columns
    from scipy.sparse import csc_matrix, dia_matrix, lil_matrix
    n = 5 * N
    offsets = [o + d for o in range(-n + N, n - N + 1, N) for d in (-1, 0, 1)]
    pat = lil_matrix(dia_matrix((np.ones((len(offsets), n)), offsets), shape=(n, n)))
    pat[:2 * N, 4 * N:] = 0
    return np.ascontiguousarray(group_columns(csc_matrix(pat)), dtype=np.int32)


def bdf(P, N, y0, t0, t1, first_step, rtol, atol, groups=None, t_eval=None, max_steps_out=1 << 20, max_events=4096, max_attempts=0):
    """scipy solve_ivp(method="BDF", jac_sparsity=...) restated (same conventions as radau below)."""
    return _implicit("marl_oracle_bdf", P, N, y0, t0, t1, first_step, rtol, atol, groups, t_eval, max_steps_out, max_events, max_attempts)


def radau(P, N, y0, t0, t1, first_step, rtol, atol, groups=None, t_eval=None, max_steps_out=1 << 20, max_events=4096,
          max_attempts=0):
    """scipy solve_ivp(method="Radau", jac_sparsity=...) restated.  Returns (y_final, stats, step_times, y_eval, t_events);
    stats.njev / stats.nlu as scipy counts them."""
    return _implicit("marl_oracle_radau", P, N, y0, t0, t1, first_step, rtol, atol, groups, t_eval, max_steps_out, max_events, max_attempts)


def _implicit(entry, P, N, y0, t0, t1, first_step, rtol, atol, groups, t_eval, max_steps_out, max_events, max_attempts):
    y = np.array(y0, dtype=np.float64)
    st = MarlStats()
    te = None if t_eval is None else np.ascontiguousarray(t_eval, dtype=np.float64)
    n_eval = 0 if te is None else te.size
    y_eval = np.empty((max(n_eval, 1), y.size))
    steps = np.empty(max_steps_out)
    nsteps = C.c_int64(0)
    tev = np.full((NEVENTS, max_events), np.nan)
    g = None if groups is None else np.ascontiguousarray(groups, dtype=np.int32)
    getattr(lib(), entry)(C.byref(P), N, _ptr(y), t0, t1, first_step, rtol, atol, _ptr(g) if g is not None else None,
                            _ptr(te) if n_eval else None, n_eval, _ptr(y_eval), _ptr(steps), max_steps_out,
                            C.byref(nsteps), _ptr(tev), max_events, max_attempts, C.byref(st))
    t_events = [tev[e, :min(int(st.n_events[e]), max_events)].copy() for e in range(NEVENTS)]
    return y, st, steps[:min(nsteps.value, max_steps_out)].copy(), y_eval[:n_eval], t_events
