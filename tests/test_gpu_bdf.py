"""The implicit BDF path on the GPU (marl_integrate_bdf: scipy's BDF step logic on the host, RHS / finite-difference Jacobian /
block-tridiagonal factorisation of I - c J by cyclic reduction / every vector operation on the device) against scipy BDF driving the
REFERENCE's RHS with the reference's jac_sparsity (goldens bdf_traj_*.npz, oracle/make_goldens.py gen_bdf) and against the CPU oracle.

As for Radau (tests/test_gpu_radau.py): where every decision matches scipy's, the states agree closely; one Newton iteration more or
less at a knife-edge convergence test changes the following step sizes and the solutions then agree like two correct runs do."""
import json

import numpy as np
import pytest

from common import GOLDEN, record_parity

pytestmark = pytest.mark.gpu
STATE_TOL = 5e-6


def _model(name):
    from dataclasses import asdict
    from marlpde_amd.LHeureux_model import LMAHeureuxPorosityDiff
    from marlpde_amd.parameters import Map_Scenario
    g = np.load(f"{GOLDEN}/bdf_traj_{name}.npz")
    N = int(g["N"])
    p = asdict(Map_Scenario()) | json.loads(str(g["overrides"])) | {"N": N}
    return g, p, LMAHeureuxPorosityDiff.from_scenario(p, device=0)


def _close_counts(res, nfev, njev, nlu, steps):
    return abs(res.nfev - nfev) <= max(6, 0.03 * nfev) and abs(res.njev - njev) <= 2 and abs(res.nlu - nlu) <= 4 and abs(res.n_accepted - steps) <= 3


@pytest.mark.parametrize("groups", ["scipy", None])
@pytest.mark.parametrize("name", ["A", "matlab", "A_N64_tight"])
def test_bdf_reproduces_scipy_on_the_reference_rhs(oracle, name, groups):
    g, p, eq = _model(name)
    N = int(g["N"])
    grp = oracle.scipy_groups(N) if groups == "scipy" else None
    res = eq.integrate_bdf(g["y0"], tuple(g["t_span"]), float(g["first_step"]), float(g["rtol"]), float(g["atol"]), t_eval=g["t_span"], groups=grp)
    eq.close()
    steps = len(g["step_times"]) - 1
    print(name, groups, (res.nfev, res.njev, res.nlu, res.n_accepted), "scipy", (int(g["nfev"]), int(g["njev"]), int(g["nlu"]), steps))
    assert res.status == 0
    assert _close_counts(res, int(g["nfev"]), int(g["njev"]), int(g["nlu"]), steps)
    want = (int(g["nfev"]), int(g["njev"]), int(g["nlu"]), steps)
    same = (res.nfev, res.njev, res.nlu, res.n_accepted) == want
    record_parity("bdf", f"{name}|groups={groups}", (res.nfev, res.njev, res.nlu, res.n_accepted), want, same)
    # all six combinations take every decision scipy takes (326/10/45/120, 309/6/42/132, 490/6/40/180): asserted, so that a regression
    # to "close" cannot pass silently; the looser bounds above only document what a knife-edge flip could cost
    assert same, (name, groups)
    tol = STATE_TOL
    assert np.max(np.abs(res.y_final - g["y_final"])) <= tol
    assert np.max(np.abs(res.y[:, 0] - g["y0"])) <= 1e-9 and np.max(np.abs(res.y[:, -1] - g["y_final"])) <= tol
    if same:
        assert [len(e) for e in res.t_events] == list(g["n_events"])
        if sum(g["n_events"]):
            assert np.max(np.abs(np.concatenate(res.t_events) - g["t_events"])) <= 5e-4


def test_bdf_against_the_oracle_on_a_finer_grid(oracle):
    """N = 1000 (beyond the one-launch solve: per-level cyclic reduction), Scenario A to t = 0.05: the GPU run against the C oracle."""
    from dataclasses import asdict
    from marlpde_amd.LHeureux_model import LMAHeureuxPorosityDiff
    from marlpde_amd.parameters import Map_Scenario
    N = 1000
    p = asdict(Map_Scenario()) | {"Phi0": 0.6, "PhiIni": 0.5, "PhiNR": 0.6, "N": N}
    y0 = np.concatenate([np.full(N, p[k]) for k in ("CAIni", "CCIni", "cCaIni", "cCO3Ini", "PhiIni")])
    y, st, steps, _, _ = oracle.bdf(oracle.params_from_dict(p), N, y0, 0.0, 0.05, 1e-6, 1e-3, 1e-3)
    eq = LMAHeureuxPorosityDiff.from_scenario(p, device=0)
    res = eq.integrate_bdf(y0, (0.0, 0.05), 1e-6, 1e-3, 1e-3, events=False)
    eq.close()
    print((res.nfev, res.njev, res.nlu, res.n_accepted), (st.nfev, st.njev, st.nlu, st.n_accepted))
    assert res.status == st.status == 0
    assert _close_counts(res, st.nfev, st.njev, st.nlu, st.n_accepted)
    same = (res.nfev, res.njev, res.nlu, res.n_accepted) == (st.nfev, st.njev, st.nlu, st.n_accepted)
    assert np.max(np.abs(res.y_final - y)) <= (STATE_TOL if same else 2e-3)


def test_integrate_equations_with_bdf_goes_native(tmp_path, monkeypatch):
    """The drop-in driver with method="BDF": the native path (no scipy solver module involved), result file like the reference's."""
    import sys
    from dataclasses import asdict, replace
    from marlpde_amd.Evolve_scenario import integrate_equations
    from marlpde_amd.parameters import Map_Scenario, Solver, Tracker
    monkeypatch.chdir(tmp_path)
    for m in [m for m in sys.modules if m.startswith("scipy.integrate._ivp")]:
        del sys.modules[m]
    sp = asdict(replace(Solver(), method="BDF"))
    last, covered, depths, Xstar, folder = integrate_equations(sp, asdict(Tracker()), asdict(Map_Scenario()) | {"Phi0": 0.6, "PhiIni": 0.5, "PhiNR": 0.6})
    assert not any(m.startswith("scipy.integrate._ivp.bdf") for m in sys.modules)
    gold = np.load(f"{GOLDEN}/bdf_traj_A.npz")["y_final"].reshape(5, 200)
    np.testing.assert_allclose(last, gold, rtol=0.1, atol=0.01)


def test_bdf_breakdown_on_a_finer_grid_is_reproduced(oracle):
    """scipy's BDF breaks down on this model on finer grids at rtol = atol = 1e-3: at N = 4000 it returns status -1 ("Required step
    size is less than spacing between numbers") at t = 0.0853225 T* (688 / 53 / 134 evaluations / Jacobians / factorisations with
    scipy 1.15 driving the oracle's RHS).  The oracle and the GPU path reproduce the failure - same status, same time."""
    from dataclasses import asdict
    from marlpde_amd.LHeureux_model import LMAHeureuxPorosityDiff
    from marlpde_amd.parameters import Map_Scenario
    N = 4000
    p = asdict(Map_Scenario()) | {"Phi0": 0.6, "PhiIni": 0.5, "PhiNR": 0.6, "N": N}
    y0 = np.concatenate([np.full(N, p[k]) for k in ("CAIni", "CCIni", "cCaIni", "cCO3Ini", "PhiIni")])
    _, st, *_ = oracle.bdf(oracle.params_from_dict(p), N, y0, 0.0, 1.0, 1e-6, 1e-3, 1e-3)
    eq = LMAHeureuxPorosityDiff.from_scenario(p, device=0)
    res = eq.integrate_bdf(y0, (0.0, 1.0), 1e-6, 1e-3, 1e-3, events=False)
    eq.close()
    print((res.status, res.t_reached, res.nfev, res.njev, res.nlu), (st.status, st.t, st.nfev, st.njev, st.nlu))
    assert res.status == st.status == -1
    assert abs(res.t_reached - 0.0853225) <= 1e-5 and abs(st.t - 0.0853225) <= 1e-5
    assert abs(res.nfev - st.nfev) <= 0.05 * st.nfev


@pytest.mark.parametrize("method", ["integrate_radau", "integrate_bdf"])
def test_zero_copy_result_words_change_nothing(method):
    """The single-instance implicit drivers read their per-iteration scalars (a sum of squares, the non-finite flag, the monitors
    record) from coherent host memory that the kernels write and the host polls (option implicit_zero_copy, default 1) instead of a
    copy + stream synchronisation per Newton iteration: same kernels, same numbers - the runs must be bit-identical."""
    g, p, eq = _model("A")
    out = []
    for zc in (0, 1):
        eq.set_option("implicit_zero_copy", zc)
        out.append(getattr(eq, method)(g["y0"], tuple(g["t_span"]), float(g["first_step"]), float(g["rtol"]), float(g["atol"])))
    eq.close()
    a, b = out
    assert (a.status, a.nfev, a.njev, a.nlu, a.n_accepted) == (b.status, b.nfev, b.njev, b.nlu, b.n_accepted)
    assert np.array_equal(a.y_final, b.y_final)
    assert all(np.array_equal(x, y) for x, y in zip(a.t_events, b.t_events))


def test_bdf_fused_newton_launch_is_bit_identical():
    """Small systems (5 N <= 2048): right-hand side, every cyclic-reduction level, update and norm of a Newton iteration in ONE launch
    (bdf::newton_fused_kernel) against the four separate kernels (option radau_fused_solve = 0): the same arithmetic in the same order."""
    g, p, eq = _model("A")
    out = []
    for fused in (0, 1):
        eq.set_option("radau_fused_solve", fused)
        out.append(eq.integrate_bdf(g["y0"], tuple(g["t_span"]), float(g["first_step"]), float(g["rtol"]), float(g["atol"])))
    eq.close()
    a, b = out
    assert (a.status, a.nfev, a.njev, a.nlu, a.n_accepted) == (b.status, b.nfev, b.njev, b.nlu, b.n_accepted)
    assert np.array_equal(a.y_final, b.y_final)


@pytest.mark.parametrize("name", ["A", "matlab", "A_N64_tight"])
def test_bdf_workgroup_solve_is_bit_identical_to_the_host_driven_newton_loop(name):
    """Small grids run solve_bdf_system (bdf.py:36-68) as ONE launch of one workgroup - predictor, every Newton iteration with its
    convergence tests evaluated on the device, the converged state's error sum and monitors (bdf::solve_wg_kernel; option bdf_solve_wg,
    default on) - instead of an RHS launch, a linear-algebra launch and a wait per iteration with the tests on the host.  Same vector
    arithmetic, same scalar tests: statistics, final state and event roots must be identical, not close."""
    g, p, eq = _model(name)
    out = []
    for wg in (0, 1):
        eq.set_option("bdf_solve_wg", wg)
        out.append(eq.integrate_bdf(g["y0"], tuple(g["t_span"]), float(g["first_step"]), float(g["rtol"]), float(g["atol"])))
    eq.close()
    a, b = out
    assert (a.status, a.nfev, a.njev, a.nlu, a.n_accepted, a.n_rejected) == (b.status, b.nfev, b.njev, b.nlu, b.n_accepted, b.n_rejected)
    assert np.array_equal(a.y_final, b.y_final)
    assert all(np.array_equal(x, y) for x, y in zip(a.t_events, b.t_events))


def test_bdf_workgroup_solve_with_the_variable_porosity_diffusion(oracle):
    """The dPhi_variable instantiation (solve_wg_kernel<VD = true>): bit-identical to the host-driven loop, and the run agrees with the
    oracle's BDF on the same parameters (statistics equal, states to the solver's tolerance)."""
    from common import scenario
    from marlpde_amd.LHeureux_model import LMAHeureuxPorosityDiff
    N = 200
    p = scenario("A", N) | {"dPhi_variable": True}
    y0 = np.concatenate([np.full(N, p[k]) for k in ("CAIni", "CCIni", "cCaIni", "cCO3Ini", "PhiIni")])
    eq = LMAHeureuxPorosityDiff.from_scenario(p, device=0)
    out = []
    for wg in (0, 1):
        eq.set_option("bdf_solve_wg", wg)
        out.append(eq.integrate_bdf(y0, (0.0, 0.3), 1e-6, 1e-3, 1e-3))
    eq.close()
    a, b = out
    assert (a.status, a.nfev, a.njev, a.nlu, a.n_accepted, a.n_rejected) == (b.status, b.nfev, b.njev, b.nlu, b.n_accepted, b.n_rejected)
    assert np.array_equal(a.y_final, b.y_final)
    y, st, *_ = oracle.bdf(oracle.params_from_dict(p | {"dPhi_variable": 1}), N, y0, 0.0, 0.3, 1e-6, 1e-3, 1e-3)
    assert st.status == b.status == 0
    assert abs(b.nfev - st.nfev) <= max(6, 0.03 * st.nfev) and abs(b.n_accepted - st.n_accepted) <= 3
    assert np.max(np.abs(b.y_final - y)) < 2e-3


def test_bdf_cyclic_reduction_front_end_at_2048_cells():
    """BDF takes the cyclic-reduction front end for its real system I - c J from 2048 cells as Radau does: against plain PCR on a span
    the method survives (scipy's BDF breaks down on the finer grids of this model, see the breakdown test) - same decisions, states to
    far below the tolerance; and the tail launches change no bit."""
    from common import scenario
    from marlpde_amd.LHeureux_model import LMAHeureuxPorosityDiff
    N = 2048
    p = scenario("A", N)
    y0 = np.concatenate([np.full(N, p[k]) for k in ("CAIni", "CCIni", "cCaIni", "cCO3Ini", "PhiIni")])
    eq = LMAHeureuxPorosityDiff.from_scenario(p, device=0)
    out = []
    for cr, tail in ((0, 1), (-1, 1), (-1, 0)):
        eq.set_option("radau_cr", cr)
        eq.set_option("radau_cr_tail", tail)
        out.append(eq.integrate_bdf(y0, (0.0, 0.02), 1e-6, 1e-3, 1e-3))
    eq.close()
    a, b, c = out
    print([(r.status, r.nfev, r.njev, r.nlu, r.n_accepted) for r in out], float(np.max(np.abs(a.y_final - b.y_final))))
    assert a.status == b.status == c.status == 0
    assert (b.nfev, b.njev, b.nlu, b.n_accepted) == (c.nfev, c.njev, c.nlu, c.n_accepted) and np.array_equal(b.y_final, c.y_final)
    assert abs(a.nfev - b.nfev) <= max(6, 0.03 * a.nfev) and abs(a.n_accepted - b.n_accepted) <= 3
    assert np.max(np.abs(a.y_final - b.y_final)) < 1e-3


def test_run_sweep_bdf_driver_equals_single_runs():
    """marlpde_amd.sweep.run_sweep_bdf: a shard's instances as concurrent single BDF runs (one context and stream per worker thread):
    every instance bit for bit its single run, whatever the number of workers."""
    import time
    from dataclasses import asdict
    from marlpde_amd.LHeureux_model import LMAHeureuxPorosityDiff
    from marlpde_amd.parameters import Map_Scenario
    from marlpde_amd.sweep import HipSweepEngine, initial_states, product_grid, run_sweep_bdf
    N = 200
    base = asdict(Map_Scenario()) | {"N": N}
    insts = product_grid(Phi0=[0.55, 0.6, 0.65], k3=[0.02, 0.05, 0.08])
    for d in insts:
        d.update(PhiIni=d["Phi0"], PhiNR=d["Phi0"], k4=d["k3"])
    y0 = initial_states(base, insts)
    t0 = time.time()
    y, status, acc, rej, t = run_sweep_bdf(base, insts, (0.0, 0.5), 1e-6, 1e-3, 1e-3, device=0)
    t_sweep = time.time() - t0
    ref = []
    t0 = time.time()
    for b, d in enumerate(insts):
        one = LMAHeureuxPorosityDiff.from_scenario(base | d, device=0)
        ref.append(one.integrate_bdf(y0[b], (0.0, 0.5), 1e-6, 1e-3, 1e-3))
        one.close()
    t_serial = time.time() - t0
    print(f"9 instances: threaded sweep {t_sweep:.3f} s, one after another {t_serial:.3f} s")
    assert list(status) == [r.status for r in ref] and list(acc) == [r.n_accepted for r in ref] and list(rej) == [r.n_rejected for r in ref]
    assert np.array_equal(y, np.stack([r.y_final for r in ref]))
    eng = HipSweepEngine(base, insts, 0)
    y1, res1 = eng.integrate_bdf(y0, (0.0, 0.5), 1e-6, 1e-3, 1e-3, 0, workers=1)
    eng.close()
    assert np.array_equal(y1, y) and [r.nfev for r in res1] == [r.nfev for r in ref]


def test_set_scenario_equals_a_fresh_model():
    """marl_ctx_set_params / LMAHeureuxPorosityDiff.set_scenario: after taking another scenario's parameters an existing model gives the
    bits a freshly constructed one gives - derived constants, RHS, an RK45 run and a Radau run - also when the dPhi switch changes."""
    from common import scenario
    from marlpde_amd.LHeureux_model import LMAHeureuxPorosityDiff
    N = 200
    pa, pb = scenario("A", N), scenario("matlab", N) | {"dPhi_variable": True}
    #  scenarios whose COLUMN differs: another depth scale Xstar (with Tstar = Xstar / sedimentation rate) and another max_depth -
    # the grid length max_depth / Xstar, delta_x, the cell centres and the depth masks must follow, as in a fresh model
    pc = scenario("A", N) | {"Xstar": 1000.0, "Tstar": 10000.0}
    pd = scenario("default", N) | {"max_depth": 400.0}
    rng = np.random.default_rng(3)
    eq = LMAHeureuxPorosityDiff.from_scenario(pa, device=0)
    for p in (pb, pc, pd, pa):
        eq.set_scenario(p)
        fresh = LMAHeureuxPorosityDiff.from_scenario(p, device=0)
        y0 = np.concatenate([np.full(N, p[k]) for k in ("CAIni", "CCIni", "cCaIni", "cCO3Ini", "PhiIni")]) * (1 + 0.01 * rng.standard_normal(5 * N))
        assert eq.Depths.length == fresh.Depths.length == p["max_depth"] / p["Xstar"]
        assert np.array_equal(eq.Depths.axes_coords[0], fresh.Depths.axes_coords[0])
        assert np.array_equal(eq.not_too_shallow, fresh.not_too_shallow) and np.array_equal(eq.not_too_deep, fresh.not_too_deep)
        for name in ("Da", "presum", "rhorat", "dPhi_fixed", "mask_lo", "mask_hi", "delta_x"):
            assert getattr(eq, name) == getattr(fresh, name), name
        assert np.array_equal(eq.fun(0.0, y0), fresh.fun(0.0, y0))
        a, b = eq.integrate_rk45(y0, (0.0, 2e-4), 1e-6, 1e-3, 1e-3), fresh.integrate_rk45(y0, (0.0, 2e-4), 1e-6, 1e-3, 1e-3)
        assert (a.n_accepted, a.n_rejected) == (b.n_accepted, b.n_rejected) and np.array_equal(a.y[:, -1], b.y[:, -1])
        a, b = eq.integrate_radau(y0, (0.0, 0.05), 1e-6, 1e-3, 1e-3), fresh.integrate_radau(y0, (0.0, 0.05), 1e-6, 1e-3, 1e-3)
        assert (a.nfev, a.njev, a.nlu) == (b.nfev, b.njev, b.nlu) and np.array_equal(a.y_final, b.y_final)
        fresh.close()
    eq.close()
