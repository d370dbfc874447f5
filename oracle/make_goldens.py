#!/usr/bin/env python3
"""Generate the committed golden vectors under tests/golden/ from the REAL reference.

TEST INFRASTRUCTURE.  Runs only in the build container (needs /root/reference); nothing on
the GPU box or in the product path imports this file.  The reference's modules are imported
UNMODIFIED from /root/reference with the stand-ins in oracle/stubs/ (numba, pint, h5py, pde)
placed first on sys.path; no reference source is copied - only inputs and outputs (data) are
written.

Groups written (all float64, little endian, .npz / .json):

  params_default.json     asdict(Map_Scenario())           <- marlpde/parameters.py:50-148
  derived_constants.json  ctor-derived constants           <- marlpde/LHeureux_model.py:36-72,130-133
  rhs_vectors.npz         y -> rate via fun_numba (== pde_rhs) and via fun, 7 event values
                                                           <- LHeureux_model.py:162-288,290-522,524-593
  rk45_traj_*.npz         scipy solve_ivp(RK45) on the reference's fun_numba: accepted step
                          times, final state, nfev          <- marlpde/Evolve_scenario.py:104-109
  rk45_event_*.npz        the same with a monitor that fires: scipy's t_events (root times)
  radau_traj_*.npz        scipy solve_ivp(Radau, jac_sparsity=27 diagonals) on the reference's fun_numba - the
                          reference's default solver: step times, nfev/njev/nlu, t_events, final state
                                                           <- marlpde/parameters.py:150-199,213; Evolve_scenario.py:104-109
  ref_final_*.npy         last frame (5,200) of the reference's own HDF5 goldens and the
                          Matlab profile (5,201)            <- tests/Regression_test/data/*
  stub_pin_report.json    reference integrate_equations (Radau) through the stubs vs those
                          HDF5 goldens (this is what pins the py-pde restatement)

Usage:  python oracle/make_goldens.py [--skip-slow]
"""
import argparse
import contextlib
import io
import json
import os
import subprocess
import sys
import tempfile
import time
from dataclasses import asdict, replace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
REFDATA = os.path.join(REF, "tests", "Regression_test", "data")

sys.path[:0] = [os.path.join(HERE, "stubs"), REF]

import marlpde  # noqa: E402  (its __init__ appends marlpde/ to sys.path)
from marlpde.parameters import Map_Scenario, Solver, Tracker  # noqa: E402
from marlpde.LHeureux_model import LMAHeureuxPorosityDiff  # noqa: E402
from marlpde.Evolve_scenario import integrate_equations  # noqa: E402
from pde import CartesianGrid, ScalarField  # noqa: E402  (the stub)
from scipy.integrate import solve_ivp  # noqa: E402
import inspect  # noqa: E402

SCENARIOS = {
    "default": {},
    "A": {"Phi0": 0.6, "PhiIni": 0.5, "PhiNR": 0.6},
    "matlab": {"Phi0": 0.5, "PhiIni": 0.5, "PhiNR": 0.5, "k3": 0.01, "k4": 0.01},
    # compaction coefficient x50 -> dPhi_fixed /50 -> |Pe_Phi| > 100 on coarse grids
    "stiffphi": {"b": 0.0005 * 50},
}


class _Bar:
    def update(self, n):
        pass


def build_model(overrides, N, fv):
    """Exactly the construction of marlpde/Evolve_scenario.py:27-68, without the driver."""
    p = asdict(Map_Scenario()) | overrides | {"N": N, "FV_switch": fv}
    depths = CartesianGrid([[0, p["max_depth"] / p["Xstar"]]], [N], periodic=False)
    shallow = ScalarField.from_expression(depths, f"heaviside(x-{p['ShallowLimit'] / p['Xstar']}, 0)")
    deep = ScalarField.from_expression(depths, f"heaviside({p['DeepLimit'] / p['Xstar']}-x, 0)")
    names = [q.name for q in inspect.signature(LMAHeureuxPorosityDiff).parameters.values()]
    filt = {k: v for k, v in p.items() if k in names}
    sl = [slice(i * N, (i + 1) * N) for i in range(5)]
    eq = LMAHeureuxPorosityDiff(depths, sl, shallow, deep, **filt)
    y0 = np.concatenate([np.full(N, p[k]) for k in ("CAIni", "CCIni", "cCaIni", "cCO3Ini", "PhiIni")])
    return eq, y0, p, depths


def make_state(kind, y0, N, x, L):
    y = y0.reshape(5, N).copy()
    if kind == "uniform":
        pass
    elif kind == "noisy":
        y *= 1.0 + 0.05 * np.random.default_rng(0).standard_normal((5, N))
    elif kind == "wavy":
        for f in range(5):
            y[f] *= 1.0 + 0.2 * np.sin(2 * np.pi * 3 * x / L + 0.7 * f)
    elif kind == "oversat":
        y[2] = 1.5 * (1.0 + 0.25 * np.sin(2 * np.pi * 2 * x / L))
        y[3] = 1.5 * (1.0 + 0.25 * np.cos(2 * np.pi * 5 * x / L))
        y[4] *= 1.0 + 0.1 * np.sin(2 * np.pi * 4 * x / L + 0.3)
    else:
        raise ValueError(kind)
    return y.ravel()


def rhs_both(eq, y):
    bar = _Bar()
    eq.last_t = 0.0
    r_loop = eq.fun_numba(0.0, y, bar, 1e-5, 0.0)
    eq.last_t = 0.0
    with np.errstate(all="ignore"):
        r_np = eq.fun(0.0, y, bar, 1e-5, 0.0)
    ev = np.array([g(0.0, y, bar, 1e-5, 0.0) for g in (
        eq.zeros, eq.zeros_CA, eq.zeros_CC, eq.ones_CA_plus_CC, eq.ones_Phi, eq.zeros_U, eq.zeros_W)])
    return np.asarray(r_loop), np.asarray(r_np), ev


def gen_params():
    p = asdict(Map_Scenario())
    with open(os.path.join(OUT, "params_default.json"), "w") as fh:
        json.dump({k: (int(v) if isinstance(v, (int, np.integer)) else float(v)) for k, v in p.items()},
                  fh, indent=1, sort_keys=True)
    derived = {}
    for name, ov in SCENARIOS.items():
        eq, _, _, _ = build_model(ov, 200, 1)
        derived[name] = {k: float(getattr(eq, k)) for k in (
            "delta_x", "nu1", "nu2", "KRat", "dCa", "dCO3", "delta", "Da", "lambda_", "auxcon",
            "rhorat0", "rhorat", "presum", "F_fixed", "dPhi_fixed", "Peclet_min", "Peclet_max")}
        derived[name]["mask_cells"] = [int(i) for i in np.nonzero(eq.not_too_deep * eq.not_too_shallow)[0]]
    with open(os.path.join(OUT, "derived_constants.json"), "w") as fh:
        json.dump(derived, fh, indent=1, sort_keys=True)
    print("params: ok")


def gen_rhs():
    store = {}
    index = []
    worst = 0.0
    cases = [(s, st, fv, N) for s in SCENARIOS for st in ("uniform", "noisy", "wavy", "oversat")
             for fv in (1, 0) for N in (16, 64, 200)]
    cases += [(s, st, 1, 1024) for s in ("default", "A") for st in ("noisy", "wavy")]
    for s, st, fv, N in cases:
        eq, y0, p, depths = build_model(SCENARIOS[s], N, fv)
        x = depths._axes_coords[0]
        y = make_state(st, y0, N, x, p["max_depth"] / p["Xstar"])
        r_loop, r_np, ev = rhs_both(eq, y)
        key = f"{s}|{st}|fv{fv}|N{N}"
        scale = np.max(np.abs(r_loop.reshape(5, N)), axis=1, keepdims=True)
        d = np.max(np.abs(r_loop - r_np).reshape(5, N) / np.where(scale > 0, scale, 1.0))
        worst = max(worst, d if np.isfinite(d) else 0.0)
        store[key + "|y"] = y
        store[key + "|rate"] = r_loop
        store[key + "|rate_numpy"] = r_np
        store[key + "|events"] = ev
        index.append(key)
    store["index"] = np.array(index)
    np.savez_compressed(os.path.join(OUT, "rhs_vectors.npz"), **store)
    print(f"rhs: {len(index)} cases, loop-vs-numpy path max rel-to-max diff {worst:.2e}")


def gen_rk45(skip_slow):
    runs = [("A", 200, 2e-3, 1e-3, 1e-6, None),
            ("default", 200, 5e-4, 1e-3, 1e-6, None),
            ("A", 64, 2e-2, 1e-6, 1e-6, 5)]
    if skip_slow:
        runs = runs[2:]
    for s, N, t1, tol, h0, n_eval in runs:
        eq, y0, p, _ = build_model(SCENARIOS[s], N, 1)
        eq.last_t = 0.0
        t_eval = None if n_eval is None else np.linspace(0.0, t1, n_eval)
        t_start = time.time()
        sol = solve_ivp(eq.fun_numba, (0.0, t1), y0, method="RK45", first_step=h0, rtol=tol, atol=tol,
                        t_eval=t_eval, dense_output=True, args=[_Bar(), t1 / 1000, 0.0],
                        events=[eq.zeros, eq.zeros_CA, eq.zeros_CC, eq.ones_CA_plus_CC, eq.ones_Phi,
                                eq.zeros_U, eq.zeros_W])
        # accepted step times: dense_output keeps every accepted step boundary
        ts = np.asarray(sol.sol.ts)
        name = f"rk45_traj_{s}_N{N}.npz"
        np.savez_compressed(os.path.join(OUT, name), y0=y0, t_span=np.array([0.0, t1]), rtol=tol, atol=tol,
                            first_step=h0, step_times=ts, y_final=sol.sol(t1), nfev=sol.nfev,
                            status=sol.status, t_eval=(np.array([]) if t_eval is None else sol.t),
                            y_eval=(np.array([]) if t_eval is None else sol.y),
                            n_events=np.array([len(e) for e in sol.t_events]))
        print(f"rk45 {name}: {len(ts) - 1} accepted steps, nfev {sol.nfev}, status {sol.status}, "
              f"{time.time() - t_start:.1f}s")


EVENTS = ("zeros", "zeros_CA", "zeros_CC", "ones_CA_plus_CC", "ones_Phi", "zeros_U", "zeros_W")


def _pack_events(t_events):
    """ragged list of 7 arrays -> (concatenated times, counts)"""
    return np.concatenate([np.asarray(e, dtype=float) for e in t_events]), np.array([len(e) for e in t_events])


def gen_rk45_event():
    """scipy RK45 on the reference RHS from a state in which a monitor FIRES: a porosity dip makes min(U) negative at
    t0, the reaction term lifts the porosity and zeros_U crosses zero (marlpde/Evolve_scenario.py:104-109, 118-145:
    the root times are what the reference prints and stores)."""
    N = 400
    eq, y0, p, depths = build_model(SCENARIOS["default"], N, 1)
    L = p["max_depth"] / p["Xstar"]
    x = depths._axes_coords[0]
    y = y0.reshape(5, N).copy()
    y[4] = 0.8 - 0.04 * np.exp(-((x - 0.5 * L) / (0.08 * L)) ** 2)
    y = y.ravel()
    dx2 = (L / N) ** 2
    t1 = 1000 * dx2
    eq.last_t = 0.0
    t_start = time.time()
    sol = solve_ivp(eq.fun_numba, (0.0, t1), y, method="RK45", first_step=0.5 * dx2, rtol=1e-5, atol=1e-7,
                    dense_output=True, args=[_Bar(), t1 / 1000, 0.0],
                    events=[getattr(eq, e) for e in EVENTS])
    tev, nev = _pack_events(sol.t_events)
    np.savez_compressed(os.path.join(OUT, "rk45_event_default_N400.npz"), y0=y, t_span=np.array([0.0, t1]), rtol=1e-5, atol=1e-7,
                        first_step=0.5 * dx2, step_times=np.asarray(sol.sol.ts), y_final=sol.y[:, -1], nfev=sol.nfev,
                        status=sol.status, t_events=tev, n_events=nev)
    print(f"rk45 event golden: {len(sol.sol.ts) - 1} accepted steps, nfev {sol.nfev}, events {nev.tolist()}, "
          f"roots {tev.tolist()}, {time.time() - t_start:.1f}s")


def gen_radau(skip_slow):
    """The reference's DEFAULT solver: scipy Radau with its 27-diagonal jac_sparsity on the reference RHS, called with
    exactly the keywords of marlpde/Evolve_scenario.py:104-109 (+ dense_output=True to read the accepted step
    times; it does not change the stepping).  Cases = the three of tests/Regression_test/test_regression.py plus one
    short tight-tolerance run."""
    cases = [("A", SCENARIOS["A"], 200, {}, (0, 1)),
             ("matlab", SCENARIOS["matlab"], 200, {}, (0, 1)),
             ("A_N64_tight", SCENARIOS["A"], 64, {"rtol": 1e-6, "atol": 1e-8}, (0, 0.02))]
    if not skip_slow:
        cases.append(("high_porosity", {"Phi0": 0.8, "PhiIni": 0.8, "PhiNR": 0.8}, 200, {"first_step": 5e-7}, (0, 1)))
    from marlpde.parameters import jacobian_sparsity
    for name, ov, N, sov, span in cases:
        eq, y0, p, _ = build_model(ov, N, 1)
        eq.last_t = 0.0
        sp = asdict(Solver()) | sov | {"t_span": span}
        sp.pop("backend", None)
        for k in ("lband", "uband"):
            sp.pop(k, None)
        sp["jac_sparsity"] = jacobian_sparsity() if N == 200 else _sparsity(N)
        t_start = time.time()
        sp["dense_output"] = True
        with np.errstate(all="ignore"):
            sol = solve_ivp(eq.fun_numba, y0=y0, **sp, t_eval=np.array(span, dtype=float),
                            events=[getattr(eq, e) for e in EVENTS], args=[_Bar(), (span[1] - span[0]) / 100000, span[0]])
        tev, nev = _pack_events(sol.t_events)
        np.savez_compressed(os.path.join(OUT, f"radau_traj_{name}.npz"), y0=y0, N=N, t_span=np.array(span, dtype=float),
                            rtol=sp["rtol"], atol=sp["atol"], first_step=sp["first_step"], step_times=np.asarray(sol.sol.ts),
                            y_final=sol.y[:, -1], nfev=sol.nfev, njev=sol.njev, nlu=sol.nlu, status=sol.status,
                            t_events=tev, n_events=nev, overrides=json.dumps(ov))
        print(f"radau {name}: {len(sol.sol.ts) - 1} steps, nfev {sol.nfev} njev {sol.njev} nlu {sol.nlu} status {sol.status} "
              f"events {nev.tolist()} {time.time() - t_start:.1f}s")


def gen_bdf():
    """The other implicit method the reference's Solver names for its jac_sparsity (marlpde/parameters.py:205-219): scipy BDF on the
    reference RHS, called like gen_radau calls Radau.  Cases: Scenario A and the Matlab cross-check case at the reference's N = 200,
    one short tight-tolerance run."""
    cases = [("A", SCENARIOS["A"], 200, {}, (0, 1)),
             ("matlab", SCENARIOS["matlab"], 200, {}, (0, 1)),
             ("A_N64_tight", SCENARIOS["A"], 64, {"rtol": 1e-6, "atol": 1e-8}, (0, 0.02))]
    from marlpde.parameters import jacobian_sparsity
    for name, ov, N, sov, span in cases:
        eq, y0, p, _ = build_model(ov, N, 1)
        eq.last_t = 0.0
        sp = asdict(Solver()) | sov | {"t_span": span, "method": "BDF"}
        sp.pop("backend", None)
        for k in ("lband", "uband"):
            sp.pop(k, None)
        sp["jac_sparsity"] = jacobian_sparsity() if N == 200 else _sparsity(N)
        t_start = time.time()
        sp["dense_output"] = True
        with np.errstate(all="ignore"):
            sol = solve_ivp(eq.fun_numba, y0=y0, **sp, t_eval=np.array(span, dtype=float),
                            events=[getattr(eq, e) for e in EVENTS], args=[_Bar(), (span[1] - span[0]) / 100000, span[0]])
        tev, nev = _pack_events(sol.t_events)
        np.savez_compressed(os.path.join(OUT, f"bdf_traj_{name}.npz"), y0=y0, N=N, t_span=np.array(span, dtype=float),
                            rtol=sp["rtol"], atol=sp["atol"], first_step=sp["first_step"], step_times=np.asarray(sol.sol.ts),
                            y_final=sol.y[:, -1], nfev=sol.nfev, njev=sol.njev, nlu=sol.nlu, status=sol.status,
                            t_events=tev, n_events=nev, overrides=json.dumps(ov))
        print(f"bdf {name}: {len(sol.sol.ts) - 1} steps, nfev {sol.nfev} njev {sol.njev} nlu {sol.nlu} status {sol.status} "
              f"events {nev.tolist()} {time.time() - t_start:.1f}s")


def _sparsity(N):
    """The reference's jacobian_sparsity() is hard-wired to the default N; the same pattern for another N
    (marlpde/parameters.py:150-199: 27 diagonals, CA/CC rows x Phi columns zeroed)."""
    from scipy.sparse import csr_matrix, dia_matrix, lil_matrix
    n = 5 * N
    offsets = [o + d for o in range(-n + N, n - N + 1, N) for d in (-1, 0, 1)]
    pat = lil_matrix(dia_matrix((np.ones((len(offsets), n)), offsets), shape=(n, n)))
    pat[:2 * N, 4 * N:] = 0
    return csr_matrix(pat)


def gen_ref_h5():
    """Reduce the reference's HDF5 fixtures to their final frames with the conda h5py."""
    code = r"""
import sys, numpy as np, h5py
src, out = sys.argv[1], sys.argv[2]
def last(name, key):
    with h5py.File(src + "/" + name, "r") as f:
        return np.array(f[key])
a = last("LMAHeureuxPorosityDiff_Phi0_0.6_PhiIni_0.5.hdf5", "data")
b = last("LMAHeureuxPorosityDiff_Phi0_PhiIni_0.8.hdf5", "data")
m = last("Matlab_output_Scenario_A_Phi0_PhiIni_0.5_k3_k4_0.01.h5", "Solutions after_T*")
np.save(out + "/ref_final_scenarioA_Phi0_0.6_PhiIni_0.5.npy", a[-1])
np.save(out + "/ref_final_high_porosity_0.8.npy", b[-1])
np.save(out + "/ref_matlab_Phi_0.5_k3_k4_0.01.npy", m[:, :, 0])
# a few intermediate frames (t = 0.1, 0.25, 0.5) of scenario A: stored by the reference, unused by its tests
np.save(out + "/ref_frames_scenarioA_t0.1_0.25_0.5.npy", a[[10, 25, 50]])
print(a.shape, b.shape, m.shape)
"""
    subprocess.run(["/opt/conda/bin/python3.9", "-W", "ignore", "-c", code, REFDATA, OUT], check=True)
    print("ref hdf5 -> npy: ok")


def gen_stub_pin(skip_slow):
    """Run the reference's own driver (Radau, its default) through the stubs against its goldens."""
    report = {}
    a = np.load(os.path.join(OUT, "ref_final_scenarioA_Phi0_0.6_PhiIni_0.5.npy"))
    m = np.load(os.path.join(OUT, "ref_matlab_Phi_0.5_k3_k4_0.01.npy"))
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "run"))
        os.chdir(os.path.join(tmp, "run"))  # the driver writes ../Results/<stamp>/ (Evolve_scenario.py:157-159)
        try:
            for name, ov, gold in (("scenarioA", SCENARIOS["A"], a), ("matlab", SCENARIOS["matlab"], m)):
                sink = io.StringIO()
                t0 = time.time()
                with contextlib.redirect_stdout(sink), contextlib.redirect_stderr(sink):
                    last, *_ = integrate_equations(asdict(Solver()), asdict(Tracker()),
                                                   asdict(Map_Scenario()) | ov)
                if name == "matlab":
                    # tests/Regression_test/test_regression.py:112,136-148
                    xs = (np.arange(200) + 0.5) * (500.0 / 200)
                    g = np.stack([np.interp(xs, np.linspace(0, 500, 201), gold[f]) for f in range(5)])
                    err = np.max(np.abs(last[:, 2:] - g[:, 2:]), axis=1)
                else:
                    err = np.max(np.abs(last - gold), axis=1)
                report[name] = {"max_abs_err_per_field": [float(e) for e in err], "seconds": time.time() - t0}
                np.save(os.path.join(OUT, f"stub_radau_final_{name}.npy"), last)
                print(f"stub pin {name}: max abs err per field {err}")
        finally:
            os.chdir(cwd)
    with open(os.path.join(OUT, "stub_pin_report.json"), "w") as fh:
        json.dump(report, fh, indent=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-slow", action="store_true")
    ap.add_argument("--only", default="", help="comma list of groups: params,rhs,h5,stubpin,rk45,rk45event,radau,bdf")
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    groups = {"params": gen_params, "rhs": gen_rhs, "h5": gen_ref_h5, "stubpin": lambda: gen_stub_pin(args.skip_slow),
              "rk45": lambda: gen_rk45(args.skip_slow), "rk45event": gen_rk45_event, "radau": lambda: gen_radau(args.skip_slow),
              "bdf": gen_bdf}
    for g in (args.only.split(",") if args.only else groups):
        groups[g]()


if __name__ == "__main__":
    main()
