"""GPU parity of the implicit path: marl_integrate_radau - scipy's Radau IIA as the reference runs it by
default (marlpde/parameters.py:213 with the jac_sparsity of :150-199), RHS / finite-difference Jacobian / block-tridiagonal LU /
vector work on the device - against
  * scipy itself on the REFERENCE RHS (goldens radau_traj_*.npz from oracle/make_goldens.py): same nfev / njev / nlu and step
    sequence on the well-conditioned cases,
  * the oracle's restatement (tests/test_oracle_radau.py pins that one),
  * the reference's three HDF5 regression files through the drop-in driver, at the reference's own tolerances - these ARE
    Radau results (tests/Regression_test/test_regression.py:43-53, 73-88, 114-148).

Tolerances: the device RHS differs from numpy's in the last bits (FMA, table-driven log/exp); the finite-difference Jacobian
divides those differences by h ~ 1e-8 |y| (absolute Jacobian noise ~1e-2), and with a fresh Jacobian Newton's convergence RATE is
that noise - so the rate tests of solve_collocation_system can fall the other way between two correct implementations: one Newton
iteration more or less (nfev +-3).  While every decision matches scipy's, states agree to ~1e-6; after a flipped decision to the
solver's own tolerance (rtol = atol), which is all any Radau run of this problem is good for.  The high-porosity run crosses the pole Phi = 1: decisions flip after ~40 steps on ANY change of
rounding (the oracle already differs from scipy there), so only its first steps, its statistics (5 %) and its final profile are
compared."""
import json

import numpy as np
import pytest

from common import GOLDEN, record_parity, scenario

pytestmark = pytest.mark.gpu

# (case, radau_solver) pairs that do NOT walk scipy's decision sequence, with the reason - every other combination must reproduce
# scipy's nfev / njev / nlu / step count EXACTLY (a regression from "same" to "within 2 rtol" cannot pass silently).
# Observed with the round-3 build (profiles/r03_implicit_parity_report.jsonl): Scenario A and the tight run are exact with BOTH linear
# solvers and both column groupings; the Matlab case is exact with the default solver (cyclic reduction).
KNOWN_DIVERGING = {
    ("matlab", 1): "block Thomas (the radau_solver = 1 cross-check, not the default): one Newton iteration at a knife-edge convergence-rate test "
                   "of solve_collocation_system takes one iteration more (393 evaluations for scipy's 390; same 19 Jacobians / 80 factorisations / 38 steps)",
}

STATE_TOL = 5e-6   # observed 1.1e-6 (Scenario A, T* = 13 190 yr, 41 steps); see the module docstring


@pytest.fixture(scope="module")
def torch_cuda_radau():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def _model(name):
    from dataclasses import asdict
    from marlpde_amd.LHeureux_model import LMAHeureuxPorosityDiff
    from marlpde_amd.parameters import Map_Scenario
    g = np.load(f"{GOLDEN}/radau_traj_{name}.npz")
    p = asdict(Map_Scenario()) | json.loads(str(g["overrides"])) | {"N": int(g["N"])}
    return g, p, LMAHeureuxPorosityDiff.from_scenario(p, device=0)


def _close_counts(res, nfev, njev, nlu, steps):
    """One Newton iteration more or less (3 evaluations) where a convergence test sits on a knife edge is the most the device
    RHS's last-bit differences may cost; Jacobian refreshes / refactorisations / steps: at most one."""
    return abs(res.nfev - nfev) <= 6 and abs(res.njev - njev) <= 1 and abs(res.nlu - nlu) <= 2 and abs(res.n_accepted - steps) <= 1


@pytest.mark.parametrize("solver", [0, 1])       # 0: block parallel cyclic reduction (default), 1: sequential block Thomas
@pytest.mark.parametrize("groups", ["scipy", None])
@pytest.mark.parametrize("name", ["A", "matlab", "A_N64_tight"])
def test_radau_reproduces_scipy_on_the_reference_rhs(oracle, name, groups, solver):
    from marlpde_amd import _abi
    g, p, eq = _model(name)
    try:
        eq.set_option("radau_solver", solver)
    except _abi.MarlError as e:   # (the block-Thomas cross-check is compiled into lab builds only since round 4: -DMARL_LAB_BLOCK_THOMAS)
        assert solver == 1 and "lab builds only" in str(e)
        eq.close()
        pytest.skip("radau_solver = 1 is not compiled into the shipped library")
    N = int(g["N"])
    grp = oracle.scipy_groups(N) if groups == "scipy" else None
    res = eq.integrate_radau(g["y0"], tuple(g["t_span"]), float(g["first_step"]), float(g["rtol"]), float(g["atol"]), t_eval=g["t_span"],
                             groups=grp)
    assert res.status == 0
    print(name, groups, solver, (res.nfev, res.njev, res.nlu, res.n_accepted), "scipy", (int(g["nfev"]), int(g["njev"]), int(g["nlu"])))
    assert _close_counts(res, int(g["nfev"]), int(g["njev"]), int(g["nlu"]), len(g["step_times"]) - 1)
    # every decision as scipy took it (observed: A + Thomas, matlab + PCR, the tight run with both): the states agree closely; one
    # Newton iteration more somewhere changes an error estimate, hence the following step sizes, hence the solution at the level
    # of the solver's own tolerance (rtol = atol = 1e-3 here) - as it does between any two correct Radau implementations
    want = (int(g["nfev"]), int(g["njev"]), int(g["nlu"]), len(g["step_times"]) - 1)
    same = (res.nfev, res.njev, res.nlu, res.n_accepted) == want
    record_parity("radau", f"{name}|groups={groups}|radau_solver={solver}", (res.nfev, res.njev, res.nlu, res.n_accepted), want, same)
    if (name, solver) not in KNOWN_DIVERGING:
        assert same, ("this combination reproduced scipy's statistics exactly when the list of known divergences was written", name, groups, solver)
    tol = STATE_TOL if same else 2 * float(g["rtol"])
    assert np.max(np.abs(res.y_final - g["y_final"])) <= tol
    assert np.array_equal(res.y[:, 0], g["y0"]) and np.max(np.abs(res.y[:, -1] - g["y_final"])) <= tol
    assert [len(e) for e in res.t_events] == list(g["n_events"])
    if sum(g["n_events"]):   # min(CA) grazes zero with slope ~2e-4: ill-conditioned root (see tests/test_oracle_radau.py)
        assert np.max(np.abs(np.concatenate(res.t_events) - g["t_events"])) <= 5e-4
    # the same run by the oracle: closer than scipy (same restated algorithm; banded LU there, block Thomas here)
    y, st, *_ = oracle.radau(oracle.params_from_dict(p), N, g["y0"], *g["t_span"], float(g["first_step"]), float(g["rtol"]), float(g["atol"]),
                             groups=grp)
    assert _close_counts(res, st.nfev, st.njev, st.nlu, st.n_accepted)
    assert np.max(np.abs(res.y_final - y)) <= (STATE_TOL if (res.nfev, res.njev, res.nlu) == (st.nfev, st.njev, st.nlu) else 2 * float(g["rtol"]))
    eq.close()


def test_radau_high_porosity_case(oracle):
    g, p, eq = _model("high_porosity")
    import time
    t0 = time.time()
    res = eq.integrate_radau(g["y0"], tuple(g["t_span"]), float(g["first_step"]), float(g["rtol"]), float(g["atol"]))
    print(f"high porosity Radau on the GPU: {time.time() - t0:.2f} s, nfev {res.nfev} njev {res.njev} nlu {res.nlu} steps {res.n_accepted}")
    assert res.status == 0
    for mine, ref in ((res.nfev, g["nfev"]), (res.njev, g["njev"]), (res.nlu, g["nlu"]), (res.n_accepted, len(g["step_times"]) - 1)):
        assert abs(mine - int(ref)) <= 0.05 * int(ref), (mine, int(ref))
    assert np.max(np.abs(res.y_final - g["y_final"])) <= 1e-4
    # porosity crosses one twice: the first crossing lies before the trajectories part, the second long after
    assert res.t_events[4].size == 2 and abs(res.t_events[4][0] - g["t_events"][0]) <= 1e-6 and abs(res.t_events[4][1] - g["t_events"][1]) <= 1e-3
    gold = np.load(f"{GOLDEN}/ref_final_high_porosity_0.8.npy")
    np.testing.assert_allclose(res.y_final.reshape(5, 200), gold, rtol=0.1, atol=0.01)
    eq.close()


@pytest.mark.parametrize("name,gold_file,first_step", [("A", "ref_final_scenarioA_Phi0_0.6_PhiIni_0.5.npy", 1e-6),
                                                        ("default", "ref_final_high_porosity_0.8.npy", 5e-7),
                                                        ("matlab", "ref_matlab_Phi_0.5_k3_k4_0.01.npy", 1e-6)])
def test_reference_regression_cases_with_the_default_solver(name, gold_file, first_step):
    """The reference's three regression tests, as written there: integrate_equations(asdict(Solver()), asdict(Tracker()),
    asdict(Map_Scenario()) | overrides) with the DEFAULT method (Radau) - no scipy in the loop here."""
    import sys
    from dataclasses import asdict, replace
    from marlpde_amd.Evolve_scenario import integrate_equations
    from marlpde_amd.parameters import Solver, Tracker
    solver = asdict(replace(Solver(), first_step=first_step))
    assert solver["method"] == "Radau" and solver["backend"] == "hip"
    gold = np.load(f"{GOLDEN}/{gold_file}")
    before = set(sys.modules)
    last, covered, *_ = integrate_equations(solver, asdict(Tracker()), scenario(name), results_root=None, verbose=False)
    assert not any(m.startswith("scipy.integrate._ivp") for m in set(sys.modules) - before), "scipy's solvers must not be in the loop"
    assert covered == pytest.approx(13190.0) and last.shape == (5, 200)
    if name == "matlab":
        xs = (np.arange(200) + 0.5) * 2.5
        interp = np.stack([np.interp(xs, np.linspace(0, 500, 201), gold[f]) for f in range(5)])
        np.testing.assert_allclose(last[:, 2:], interp[:, 2:], atol=0.05)
    else:
        np.testing.assert_allclose(last, gold, rtol=0.1, atol=0.01)
        assert np.max(np.abs(last - gold)) < 2e-3    # (the stub-hosted reference itself is within 1e-4 of these files; rtol = 1e-3)


def test_radau_api_errors_and_budget(oracle):
    from marlpde_amd._abi import MarlError
    g, p, eq = _model("A")
    with pytest.raises(MarlError, match="first_step"):
        eq.integrate_radau(g["y0"], (0.0, 1.0), 2.0, 1e-3, 1e-3)
    with pytest.raises(MarlError, match="share rows"):
        eq.integrate_radau(g["y0"], (0.0, 1.0), 1e-6, 1e-3, 1e-3, groups=np.zeros(1000, dtype=np.int32))
    res = eq.integrate_radau(g["y0"], (0.0, 1.0), 1e-6, 1e-3, 1e-3, max_attempts=5)
    y, st, *_ = oracle.radau(oracle.params_from_dict(p), 200, g["y0"], 0.0, 1.0, 1e-6, 1e-3, 1e-3, max_attempts=5)
    assert res.status == 2 == st.status and res.n_accepted == st.n_accepted and res.t_reached == pytest.approx(st.t, rel=1e-9)
    eq.close()


@pytest.mark.parametrize("method", ["radau", "bdf"])
def test_cyclic_reduction_front_end_on_the_reference_grid_walks_scipys_decisions(method):
    """Large grids of single runs put levels of block cyclic reduction in front of PCR (csrc/marl_radau_cr.h; automatic from 2048 cells).
    Forced onto the reference's N = 200 grid (3 levels: 200 -> 100 -> 50 -> 25 rows, PCR on the 25), the run must still make every
    decision scipy makes on the reference's RHS - same nfev / njev / nlu / steps as the goldens - and land on PCR's state to far below
    the solver tolerance (the rows that stay see PCR's arithmetic; the eliminated ones differ in rounding)."""
    g, p, eq = _model("A")
    run = eq.integrate_radau if method == "radau" else eq.integrate_bdf
    gold = g if method == "radau" else np.load(f"{GOLDEN}/bdf_traj_A.npz")
    out = []
    for cr in (0, 3):
        eq.set_option("radau_cr", cr)
        out.append(run(g["y0"], tuple(g["t_span"]), float(g["first_step"]), float(g["rtol"]), float(g["atol"])))
    eq.close()
    a, b = out
    assert a.status == b.status == 0
    want = (int(gold["nfev"]), int(gold["njev"]), int(gold["nlu"]), len(gold["step_times"]) - 1)
    assert (a.nfev, a.njev, a.nlu, a.n_accepted) == want
    assert (b.nfev, b.njev, b.nlu, b.n_accepted) == want
    assert np.max(np.abs(a.y_final - b.y_final)) < 2e-5        # observed 2.2e-6 (Radau), 2.2e-7 (BDF)
    for ea, eb in zip(a.t_events, b.t_events):
        assert len(ea) == len(eb) and np.allclose(ea, eb, rtol=0, atol=5e-4)    # (the tolerance of the scipy goldens; observed 1.5e-5)


def test_cyclic_reduction_front_end_against_pcr_and_the_oracle_at_4000_cells(oracle):
    """N = 4000 (automatic: 4 levels down to 250 rows, then the one-launch PCR solve): the statistics of the run equal plain PCR's and the
    oracle's (banded LU on the CPU), the states agree within the solver's tolerance."""
    from marlpde_amd.LHeureux_model import LMAHeureuxPorosityDiff
    N = 4000
    p = scenario("A", N)
    y0 = np.concatenate([np.full(N, p[k]) for k in ("CAIni", "CCIni", "cCaIni", "cCO3Ini", "PhiIni")])
    eq = LMAHeureuxPorosityDiff.from_scenario(p, device=0)
    out = []
    for cr in (0, -1):
        eq.set_option("radau_cr", cr)
        out.append(eq.integrate_radau(y0, (0.0, 1.0), 1e-6, 1e-3, 1e-3))
    eq.close()
    a, b = out
    y, st, *_ = oracle.radau(oracle.params_from_dict(p), N, y0, 0.0, 1.0, 1e-6, 1e-3, 1e-3)
    assert a.status == b.status == st.status == 0
    assert (a.nfev, a.njev, a.nlu, a.n_accepted) == (st.nfev, st.njev, st.nlu, st.n_accepted)
    assert (b.nfev, b.njev, b.nlu, b.n_accepted) == (st.nfev, st.njev, st.nlu, st.n_accepted)
    assert np.max(np.abs(b.y_final - y)) < 1e-3 and np.max(np.abs(a.y_final - y)) < 1e-3      # observed 2.2e-4, 4.6e-5 (rtol = atol = 1e-3)


@pytest.mark.parametrize("N", [2048, 5003])
def test_cyclic_reduction_tail_launches_are_bit_identical_to_level_by_level(N):
    """From the first level of at most 8192 rows on, a cyclic-reduction solve walks all remaining levels in one launch each way
    (cr_rhs_tail_kernel / cr_back_tail_kernel: tiles with recomputed edges, option radau_cr_tail, default on) instead of one launch per
    level - the same recurrences in the same order.  N = 5003: odd row counts on the way down (5003, 2501, 1250, 625, 312, 156)."""
    from marlpde_amd.LHeureux_model import LMAHeureuxPorosityDiff
    p = scenario("A", N)
    y0 = np.concatenate([np.full(N, p[k]) for k in ("CAIni", "CCIni", "cCaIni", "cCO3Ini", "PhiIni")])
    eq = LMAHeureuxPorosityDiff.from_scenario(p, device=0)
    out = []
    for tail in (0, 1):
        eq.set_option("radau_cr_tail", tail)
        out.append(eq.integrate_radau(y0, (0.0, 0.2), 1e-6, 1e-3, 1e-3))
    eq.close()
    a, b = out
    assert a.status == b.status == 0
    assert (a.nfev, a.njev, a.nlu, a.n_accepted) == (b.nfev, b.njev, b.nlu, b.n_accepted)
    assert np.array_equal(a.y_final, b.y_final)


@pytest.mark.parametrize("name", ["A", "matlab"])
def test_radau_fused_solve_is_bit_identical_to_per_level_solve(oracle, name):
    """Small systems (5 N <= 2048) run all levels of a PCR solve in one launch (pcr_solve_fused_kernel: right-hand side in LDS, a
    barrier between levels) instead of one launch per level.  The recurrences are the same and their multiply-adds are spelled
    out (left to the compiler, the complex ones were contracted differently in the two kernels - a last-bit difference that the
    ill-conditioned systems of this model amplify to 1e-6 within 20 steps): the whole integration must be bit-identical."""
    g, p, eq = _model(name)
    out = []
    for fused in (0, 1):
        eq.set_option("radau_fused_solve", fused)
        out.append(eq.integrate_radau(g["y0"], tuple(g["t_span"]), float(g["first_step"]), float(g["rtol"]), float(g["atol"])))
    eq.close()
    a, b = out
    assert a.status == b.status == 0
    assert (a.nfev, a.njev, a.nlu, a.n_accepted) == (b.nfev, b.njev, b.nlu, b.n_accepted)
    assert np.array_equal(a.y_final, b.y_final)


def test_radau_sweep_equals_instance_by_instance(torch_cuda_radau, oracle):
    """A sweep of Radau integrations (marl_sweep_radau_dev: per-instance step logic as a state machine on the device, all
    instances advanced together) against the same instances integrated one at a time: the three scenarios of the reference's
    tests plus variations of the knobs they turn (Phi0, PhiIni, k3 = k4)."""
    torch = torch_cuda_radau
    from dataclasses import asdict
    from marlpde_amd.LHeureux_model import LMAHeureuxPorosityDiff
    from marlpde_amd.parameters import Map_Scenario
    N = 200
    base = asdict(Map_Scenario()) | {"N": N}
    inst = [{"Phi0": 0.6, "PhiIni": 0.5, "PhiNR": 0.6},                                   # Scenario A
            {"Phi0": 0.5, "PhiIni": 0.5, "PhiNR": 0.5, "k3": 0.01, "k4": 0.01},         # the Matlab cross-check case
            {"Phi0": 0.7, "PhiIni": 0.6, "PhiNR": 0.6},
            {"Phi0": 0.55, "PhiIni": 0.55, "PhiNR": 0.55, "k3": 0.03, "k4": 0.03},
            {"Phi0": 0.65, "PhiIni": 0.5, "PhiNR": 0.5, "k3": 0.05, "k4": 0.05},
            {"Phi0": 0.6, "PhiIni": 0.6, "PhiNR": 0.6}]
    y0 = np.stack([np.concatenate([np.full(N, (base | d)[k]) for k in ("CAIni", "CCIni", "cCaIni", "cCO3Ini", "PhiIni")]) for d in inst])
    eq = LMAHeureuxPorosityDiff.from_scenario(base, device=0, instances=inst)
    eq.use_stream(torch.cuda.current_stream().cuda_stream)
    yd = torch.from_numpy(y0).cuda()
    res = eq.sweep_radau_device(yd.data_ptr(), (0.0, 1.0), 1e-6, 1e-3, 1e-3)
    got = yd.cpu().numpy()
    eq.close()
    for b, d in enumerate(inst):
        one = LMAHeureuxPorosityDiff.from_scenario(base | d, device=0)
        ref = one.integrate_radau(y0[b], (0.0, 1.0), 1e-6, 1e-3, 1e-3)
        one.close()
        print(b, (res[b].nfev, res[b].njev, res[b].nlu, res[b].n_accepted), (ref.nfev, ref.njev, ref.nlu, ref.n_accepted))
        assert res[b].status == 0 == ref.status and res[b].t_reached == 1.0
        # the controller's scalar arithmetic runs on the device here and on the host there (pow, sqrt of different libms): where
        # every decision still matches, the states agree closely; a scenario with ~100 steps and ~85 Jacobians leaves the common
        # path at some knife-edge test and then differs like any two correct runs do - statistics within 10 %, states within the
        # solver's tolerance
        same = (res[b].nfev, res[b].njev, res[b].nlu, res[b].n_accepted) == (ref.nfev, ref.njev, ref.nlu, ref.n_accepted)
        # (the third instance - ~100 steps, ~90 Jacobians, many rejections - is the sensitive one: other, equally valid roundings
        #  of the complex multiply-adds in the linear solves gave nfev 1612 / 1704 / 1861 / 1991, njev 84 / 86 / 96 / 107)
        for mine, theirs in ((res[b].nfev, ref.nfev), (res[b].njev, ref.njev), (res[b].nlu, ref.nlu), (res[b].n_accepted, ref.n_accepted)):
            assert abs(mine - theirs) <= max(6, 0.1 * theirs)
        if same:   # (observed up to 1.9e-5: the step sizes themselves differ in the last bits between the two controllers)
            assert np.max(np.abs(got[b] - ref.y_final)) <= 1e-4
        else:   # two correct runs at rtol = atol = 1e-3 over ~100 steps: compared like the reference compares runs (test_regression.py:29-30)
            np.testing.assert_allclose(got[b], ref.y_final, rtol=0.1, atol=0.01)
        if same:
            assert list(res[b].n_events) == [len(e) for e in ref.t_events]
    assert sum((r.nfev, r.njev, r.nlu) == (393, 23, 76) for r in res[:1]) == 1      # Scenario A: scipy's own statistics
    # the first two are the reference's regression cases: final profiles within its tolerances of its HDF5 data
    gold = np.load(f"{GOLDEN}/ref_final_scenarioA_Phi0_0.6_PhiIni_0.5.npy")
    np.testing.assert_allclose(got[0].reshape(5, N), gold, rtol=0.1, atol=0.01)


@pytest.mark.parametrize("name", ["A", "matlab", "A_N64_tight"])
def test_radau_fused_newton_launch_is_bit_identical(name):
    """Small systems (5 N <= 2048): the right-hand sides of both collocation systems, every cyclic-reduction level of both solves, the
    update and the norm of a Newton iteration in ONE launch (radau::newton_fused_kernel, option radau_fused_solve = 2; not the default: measured slower)
    against the one-launch solves (1) and the per-level launches (0): the same arithmetic in the same order.  So is the default for
    single runs (3): two launches per iteration - the two solve workgroups assemble their own right-hand sides
    (newton_solve2_kernel), the update workgroup evaluates the next iteration's stage derivatives (newton_update_rhs_kernel)."""
    g, p, eq = _model(name)
    out = []
    for fused in (0, 1, 2, 3):
        eq.set_option("radau_fused_solve", fused)
        out.append(eq.integrate_radau(g["y0"], tuple(g["t_span"]), float(g["first_step"]), float(g["rtol"]), float(g["atol"])))
    eq.close()
    for b in out[1:]:
        a = out[0]
        assert (a.status, a.nfev, a.njev, a.nlu, a.n_accepted, a.n_rejected) == (b.status, b.nfev, b.njev, b.nlu, b.n_accepted, b.n_rejected)
        assert np.array_equal(a.y_final, b.y_final)
        assert all(np.array_equal(x, y) for x, y in zip(a.t_events, b.t_events))


def test_radau_sweep_locates_event_roots_like_the_single_run(torch_cuda_radau):
    """VERDICT r2 missing #3: the reference prints and stores t_events for EVERY run (Evolve_scenario.py:118-145, 175-177); a sweep now
    locates the monitors' roots itself (marl_sweep_radau_events_dev: Brent on the accepted step's dense output as a device-side state
    machine).  Scenario A and the Matlab case inside a sweep: the same statistics as without root finding (locating roots must not
    disturb the integration), the same number of roots as sign changes, and root times equal to the single run's (which are pinned
    to scipy's by test_radau_reproduces_scipy_on_the_reference_rhs)."""
    torch = torch_cuda_radau
    from dataclasses import asdict
    from marlpde_amd.LHeureux_model import LMAHeureuxPorosityDiff
    from marlpde_amd.parameters import Map_Scenario
    N = 200
    base = asdict(Map_Scenario()) | {"N": N}
    inst = [{"Phi0": 0.6, "PhiIni": 0.5, "PhiNR": 0.6}, {"Phi0": 0.5, "PhiIni": 0.5, "PhiNR": 0.5, "k3": 0.01, "k4": 0.01},
            {"Phi0": 0.6, "PhiIni": 0.6, "PhiNR": 0.6}, {}]      # ... and the high-porosity default scenario (porosity crosses one twice, W changes sign hundreds of times)
    y0 = np.stack([np.concatenate([np.full(N, (base | d)[k]) for k in ("CAIni", "CCIni", "cCaIni", "cCO3Ini", "PhiIni")]) for d in inst])
    eq = LMAHeureuxPorosityDiff.from_scenario(base, device=0, instances=inst)
    eq.use_stream(torch.cuda.current_stream().cuda_stream)
    out = {}
    for ev in (False, True):
        yd = torch.from_numpy(y0).cuda()
        res = eq.sweep_radau_device(yd.data_ptr(), (0.0, 1.0), 1e-6, 1e-3, 1e-3, events=ev, max_events=2048)
        out[ev] = (yd.cpu().numpy(), res)
    eq.close()
    assert np.array_equal(out[False][0], out[True][0])
    for a, b in zip(out[False][1], out[True][1]):
        assert (a.status, a.nfev, a.njev, a.nlu, a.n_accepted, a.n_rejected) == (b.status, b.nfev, b.njev, b.nlu, b.n_accepted, b.n_rejected)
        assert list(a.n_events) == list(b.n_events) == [len(t) for t in b.t_events]
        assert all(np.all(np.isfinite(t)) and np.all(np.diff(t) >= 0) and np.all((t >= 0) & (t <= 1)) for t in b.t_events)
    print("roots per instance", [[len(t) for t in r.t_events] for r in out[True][1]])
    for b in (0, 1):
        one = LMAHeureuxPorosityDiff.from_scenario(base | inst[b], device=0)
        ref = one.integrate_radau(y0[b], (0.0, 1.0), 1e-6, 1e-3, 1e-3)
        one.close()
        got = out[True][1][b]
        assert (got.nfev, got.njev, got.nlu, got.n_accepted) == (ref.nfev, ref.njev, ref.nlu, ref.n_accepted)
        assert [len(t) for t in got.t_events] == [len(t) for t in ref.t_events]
        if b == 0:
            assert [len(t) for t in ref.t_events][:2] == [1, 1]      # Scenario A: the aragonite fraction reaches zero (min(y) and min(CA) cross together)
        for e in range(7):
            # (min(CA) grazes zero with slope ~2e-4 in Scenario A: the root is ill-conditioned - test_oracle_radau.py; the two controllers'
            #  step sizes differ in the last bits: sqrt / pow of two libms)
            assert np.allclose(got.t_events[e], ref.t_events[e], rtol=0, atol=5e-4), (b, e, got.t_events[e], ref.t_events[e])


def test_radau_sweep_workgroup_paths_against_the_launch_path(torch_cuda_radau):
    """Sweeps of small grids (option radau_sweep_wg): 1 = HYBRID - one persistent workgroup per instance runs the sequential
    work of the instance (step logic, Newton iterations, error estimates, accepted steps, event roots; marl_radau_wg.h) and hands it back to
    the host cycle for Jacobians and factorisations (launch kernels over work lists); 3 = the same with the Jacobian in the workgroup too
    (the default: only factorisations come back; must be bit-identical to 1); 2 = the workgroup does everything; 0 = one launch
    cycle per action (round 2).  Same step-logic function everywhere.  HYBRID must reproduce the launch path's statistics EXACTLY and its
    states to 1e-12 (same kernels for the factorisations, restated bodies with the same reduction trees for the rest); mode 2 inlines the
    factorisation into the big kernel, where the compiler contracts its complex multiply-adds differently: it agrees like two correct
    runs do (statistics within 10 %, states within the reference's tolerance; before any decision flips, to 1e-7)."""
    torch = torch_cuda_radau
    from dataclasses import asdict
    from marlpde_amd.LHeureux_model import LMAHeureuxPorosityDiff
    from marlpde_amd.parameters import Map_Scenario
    for N, t1, budget in ((200, 1.0, 0), (64, 0.3, 0), (409, 1.0, 12)):      # the reference's grid; a small one; the largest the one-workgroup solve holds
        base = asdict(Map_Scenario()) | {"N": N}
        inst = [{"Phi0": 0.6, "PhiIni": 0.5, "PhiNR": 0.6}, {"Phi0": 0.5, "PhiIni": 0.5, "PhiNR": 0.5, "k3": 0.01, "k4": 0.01},
                {"Phi0": 0.6, "PhiIni": 0.6, "PhiNR": 0.6}, {"Phi0": 0.65, "PhiIni": 0.5, "PhiNR": 0.5, "k3": 0.05, "k4": 0.05}]
        y0 = np.stack([np.concatenate([np.full(N, (base | d)[k]) for k in ("CAIni", "CCIni", "cCaIni", "cCO3Ini", "PhiIni")]) for d in inst])
        out = {}
        for wg in (0, 1, 2, 3):
            eq = LMAHeureuxPorosityDiff.from_scenario(base, device=0, instances=inst)
            eq.use_stream(torch.cuda.current_stream().cuda_stream)
            eq.set_option("radau_sweep_wg", wg)
            yd = torch.from_numpy(y0).cuda()
            res = eq.sweep_radau_device(yd.data_ptr(), (0.0, t1), 1e-6, 1e-3, 1e-3, max_attempts=budget)
            out[wg] = (yd.cpu().numpy(), res)
            eq.close()
        print(N, [[(r.nfev, r.njev, r.nlu, r.n_accepted) for r in out[m][1]] for m in (0, 1, 2)])
        for b in range(len(inst)):
            a, h, w = out[0][1][b], out[1][1][b], out[2][1][b]
            assert a.status == h.status == w.status == (2 if budget else 0)
            assert (a.nfev, a.njev, a.nlu, a.n_accepted, a.n_rejected, list(a.n_events)) == (h.nfev, h.njev, h.nlu, h.n_accepted, h.n_rejected, list(h.n_events)), (N, b)
            assert a.t_reached == h.t_reached and np.max(np.abs(out[0][0][b] - out[1][0][b])) <= 1e-12, (N, b, float(np.max(np.abs(out[0][0][b] - out[1][0][b]))))
            j = out[3][1][b]
            assert (j.status, j.nfev, j.njev, j.nlu, j.n_accepted, j.n_rejected, list(j.n_events), j.t_reached) == (h.status, h.nfev, h.njev, h.nlu, h.n_accepted, h.n_rejected, list(h.n_events), h.t_reached), (N, b)
            assert np.array_equal(out[3][0][b], out[1][0][b]), (N, b)
            for x, y in ((a.nfev, w.nfev), (a.njev, w.njev), (a.nlu, w.nlu), (a.n_accepted, w.n_accepted)):
                assert abs(x - y) <= max(6, 0.1 * x), (N, b, x, y)
            if budget:
                assert a.t_reached == pytest.approx(w.t_reached, rel=1e-6) and np.max(np.abs(out[0][0][b] - out[2][0][b])) <= 1e-7   # (observed 2e-9 / 1e-10 after 12 attempts)
            else:
                np.testing.assert_allclose(out[2][0][b], out[0][0][b], rtol=0.1, atol=0.01)


def test_radau_sweep_of_512_scenarios_properties(torch_cuda_radau):
    """A sweep the size bench.py reports (512 scenarios over Phi0 x PhiIni x k3 = k4, N = 200, to T*): every instance reaches T* with
    finite fields (positive porosity and concentrations), its statistics are in the range single runs show, and a sample of instances agrees with the same scenario
    integrated alone at the reference's tolerance (tests/Regression_test/test_regression.py:29-30)."""
    torch = torch_cuda_radau
    from dataclasses import asdict
    from marlpde_amd.LHeureux_model import LMAHeureuxPorosityDiff
    from marlpde_amd.parameters import Map_Scenario
    N, B, k = 200, 512, 8
    base = asdict(Map_Scenario()) | {"N": N}
    inst = []
    for i in range(B):
        d = {"Phi0": 0.5 + 0.2 * ((i % k) / (k - 1)), "PhiIni": 0.5 + 0.2 * (((i // k) % k) / (k - 1)), "k3": 10 ** (-2 + ((i // (k * k)) % k) / (k - 1))}
        d["PhiNR"], d["k4"] = d["PhiIni"], d["k3"]
        inst.append(d)
    y0 = np.stack([np.concatenate([np.full(N, (base | d)[q]) for q in ("CAIni", "CCIni", "cCaIni", "cCO3Ini", "PhiIni")]) for d in inst])
    eq = LMAHeureuxPorosityDiff.from_scenario(base, device=0, instances=inst)
    eq.use_stream(torch.cuda.current_stream().cuda_stream)
    yd = torch.from_numpy(y0).cuda()
    res = eq.sweep_radau_device(yd.data_ptr(), (0.0, 1.0), 1e-6, 1e-3, 1e-3)
    got = yd.cpu().numpy().reshape(B, 5, N)
    eq.close()
    assert all(r.status == 0 and r.t_reached == 1.0 for r in res)
    assert np.all(np.isfinite(got))
    print("porosity range", got[:, 4].min(), got[:, 4].max(), "solids range", got[:, :2].min(), (got[:, 0] + got[:, 1]).max())
    assert np.all(got[:, 4] > 0) and np.all(got[:, 2:4] > 0)               # porosity and solute concentrations stay positive
    nfev = np.array([r.nfev for r in res])
    assert 200 < np.median(nfev) < 600 and nfev.min() > 150
    for b in range(0, B, 73):
        one = LMAHeureuxPorosityDiff.from_scenario(base | inst[b], device=0)
        ref = one.integrate_radau(y0[b], (0.0, 1.0), 1e-6, 1e-3, 1e-3, events=False)
        one.close()
        np.testing.assert_allclose(got[b].ravel(), ref.y_final, rtol=0.1, atol=0.01)


def test_run_sweep_radau_driver_on_the_gpu(torch_cuda_radau):
    """marlpde_amd.sweep.run_sweep_radau (one rank here; shards over the ranks of a torch.distributed job, tests/test_multi_rank_cpu.py):
    the same numbers as the model's own sweep entry point."""
    torch = torch_cuda_radau
    from dataclasses import asdict
    from marlpde_amd.LHeureux_model import LMAHeureuxPorosityDiff
    from marlpde_amd.parameters import Map_Scenario
    from marlpde_amd.sweep import initial_states, product_grid, run_sweep_radau
    N = 200
    base = asdict(Map_Scenario()) | {"N": N}
    insts = product_grid(Phi0=[0.55, 0.65], k3=[0.02, 0.08])
    for d in insts:
        d.update(PhiIni=d["Phi0"], PhiNR=d["Phi0"], k4=d["k3"])
    y, status, acc, rej, t = run_sweep_radau(base, insts, (0.0, 1.0), 1e-6, 1e-3, 1e-3, device=0)
    eq = LMAHeureuxPorosityDiff.from_scenario(base, device=0, instances=insts)
    yd = torch.from_numpy(initial_states(base, insts)).cuda()
    res = eq.sweep_radau_device(yd.data_ptr(), (0.0, 1.0), 1e-6, 1e-3, 1e-3)
    eq.close()
    assert list(status) == [0] * 4 and np.all(t == 1.0)
    assert np.array_equal(y, yd.cpu().numpy()) and list(acc) == [r.n_accepted for r in res]


def _forced_small_cr(eq, levels=3):
    eq.set_option("radau_cr_small", levels)
    eq.set_option("radau_cr_small_min_n", 32)


@pytest.mark.parametrize("method", ["radau", "bdf"])
@pytest.mark.parametrize("name", ["A", "matlab", "A_N64_tight"])
def test_small_grid_cyclic_reduction_in_the_one_workgroup_solves(name, method):
    """Grids solved in one workgroup (up to 409 cells) can put up to three levels of cyclic reduction in front of PCR inside the
    one-launch solve kernels (crpcr_solve_all: the chain of levels is bound by the factor bytes that pass through one compute unit).
    Default from 205 cells; forced here onto the goldens' grids (N = 200, 64): the single runs still take every decision scipy takes
    (observed for 2 and 3 levels, Radau and BDF, all three cases - tools/lab: profiles/r03_lab_radau_wg.log), and the ways of running
    the same arithmetic - one launch per level, one launch per solve, one launch per Newton iteration / per solve_bdf_system - agree
    bit for bit."""
    g, p, eq = _model(name)
    gold = g if method == "radau" else np.load(f"{GOLDEN}/bdf_traj_{name}.npz")
    _forced_small_cr(eq)
    run = eq.integrate_radau if method == "radau" else eq.integrate_bdf
    out = []
    for fused, wg in ((1, 1), (0, 1), (2, 1), (3, 1), (1, 0)):
        eq.set_option("radau_fused_solve", fused)
        eq.set_option("bdf_solve_wg", wg)
        out.append(run(g["y0"], tuple(g["t_span"]), float(g["first_step"]), float(g["rtol"]), float(g["atol"])))
    eq.close()
    want = (int(gold["nfev"]), int(gold["njev"]), int(gold["nlu"]), len(gold["step_times"]) - 1)
    for r in out:
        assert r.status == 0 and (r.nfev, r.njev, r.nlu, r.n_accepted) == want
        assert np.array_equal(r.y_final, out[0].y_final)
        assert all(np.array_equal(x, y) for x, y in zip(r.t_events, out[0].t_events))
    assert np.max(np.abs(out[0].y_final - gold["y_final"])) <= 4 * STATE_TOL


def test_small_grid_cyclic_reduction_at_409_cells_by_default():
    """N = 409 (two unknowns per thread of the one-workgroup chain): cyclic reduction is on by default there; per-level launches, one
    launch per solve and one launch per Newton iteration give the same bits, and plain PCR (radau_cr_small = 0) the same decisions."""
    from marlpde_amd.LHeureux_model import LMAHeureuxPorosityDiff
    N = 409
    p = scenario("A", N)
    y0 = np.concatenate([np.full(N, p[k]) for k in ("CAIni", "CCIni", "cCaIni", "cCO3Ini", "PhiIni")])
    eq = LMAHeureuxPorosityDiff.from_scenario(p, device=0)
    out = []
    for fused, crs in ((1, 3), (0, 3), (2, 3), (3, 3), (1, 0)):
        eq.set_option("radau_fused_solve", fused)
        eq.set_option("radau_cr_small", crs)
        out.append(eq.integrate_radau(y0, (0.0, 1.0), 1e-6, 1e-3, 1e-3))
    eq.close()
    a, b, c, d, pcr = out
    assert a.status == pcr.status == 0
    for r in (b, c, d):
        assert (r.nfev, r.njev, r.nlu, r.n_accepted) == (a.nfev, a.njev, a.nlu, a.n_accepted) and np.array_equal(r.y_final, a.y_final)
    assert (pcr.nfev, pcr.njev, pcr.nlu, pcr.n_accepted) == (a.nfev, a.njev, a.nlu, a.n_accepted)      # 438 / 22 / 78 / 47 both ways
    assert np.max(np.abs(pcr.y_final - a.y_final)) < 1e-4


def test_small_grid_cyclic_reduction_in_sweeps(torch_cuda_radau):
    """The same levels in a sweep (batched cr_init / cr_reduce launches over the factorisation work list, crpcr_solve_all in the
    per-instance workgroups and in the batched one-launch solves): the launch-per-action cycle and the two hybrid modes give identical
    results; against single runs with the same setting the instances agree like two correct runs do (a knife-edge Newton test may fall
    the other way - why N = 200 keeps plain PCR by default)."""
    torch = torch_cuda_radau
    from dataclasses import asdict
    from marlpde_amd.LHeureux_model import LMAHeureuxPorosityDiff
    from marlpde_amd.parameters import Map_Scenario
    N = 200
    base = asdict(Map_Scenario()) | {"N": N}
    inst = [{"Phi0": 0.6, "PhiIni": 0.5, "PhiNR": 0.6}, {"Phi0": 0.5, "PhiIni": 0.5, "PhiNR": 0.5, "k3": 0.01, "k4": 0.01},
            {"Phi0": 0.6, "PhiIni": 0.6, "PhiNR": 0.6}, {"Phi0": 0.65, "PhiIni": 0.5, "PhiNR": 0.5, "k3": 0.05, "k4": 0.05}]
    y0 = np.stack([np.concatenate([np.full(N, (base | d)[k]) for k in ("CAIni", "CCIni", "cCaIni", "cCO3Ini", "PhiIni")]) for d in inst])
    out = {}
    for wg in (0, 1, 3):
        eq = LMAHeureuxPorosityDiff.from_scenario(base, device=0, instances=inst)
        eq.use_stream(torch.cuda.current_stream().cuda_stream)
        eq.set_option("radau_sweep_wg", wg)
        _forced_small_cr(eq)
        yd = torch.from_numpy(y0).cuda()
        res = eq.sweep_radau_device(yd.data_ptr(), (0.0, 1.0), 1e-6, 1e-3, 1e-3)
        out[wg] = (yd.cpu().numpy(), res)
        eq.close()
    for b, d in enumerate(inst):
        a, h, j = out[0][1][b], out[1][1][b], out[3][1][b]
        assert a.status == h.status == j.status == 0
        assert (a.nfev, a.njev, a.nlu, a.n_accepted, a.n_rejected) == (h.nfev, h.njev, h.nlu, h.n_accepted, h.n_rejected) == (j.nfev, j.njev, j.nlu, j.n_accepted, j.n_rejected)
        assert np.max(np.abs(out[0][0][b] - out[1][0][b])) <= 1e-12 and np.array_equal(out[1][0][b], out[3][0][b])
        one = LMAHeureuxPorosityDiff.from_scenario(base | d, device=0)
        _forced_small_cr(one)
        ref = one.integrate_radau(y0[b], (0.0, 1.0), 1e-6, 1e-3, 1e-3)
        one.close()
        for x, y in ((a.nfev, ref.nfev), (a.njev, ref.njev), (a.nlu, ref.nlu), (a.n_accepted, ref.n_accepted)):
            assert abs(x - y) <= max(6, 0.1 * y), (b, x, y)
        np.testing.assert_allclose(out[0][0][b], ref.y_final, rtol=0.1, atol=0.01)


def test_radau_fused_launches_with_the_variable_porosity_diffusion(oracle):
    """The dPhi_variable instantiations of the fused single-run launches (newton_update_rhs_kernel<true>, accept_fused_kernel<true>):
    bit-identical to the four-launch iteration, and the run agrees with the oracle's Radau on the same parameters."""
    from marlpde_amd.LHeureux_model import LMAHeureuxPorosityDiff
    N = 200
    p = scenario("A", N) | {"dPhi_variable": True}
    y0 = np.concatenate([np.full(N, p[k]) for k in ("CAIni", "CCIni", "cCaIni", "cCO3Ini", "PhiIni")])
    eq = LMAHeureuxPorosityDiff.from_scenario(p, device=0)
    out = []
    for fused in (1, 3):
        eq.set_option("radau_fused_solve", fused)
        out.append(eq.integrate_radau(y0, (0.0, 0.5), 1e-6, 1e-3, 1e-3))
    eq.close()
    a, b = out
    assert (a.status, a.nfev, a.njev, a.nlu, a.n_accepted, a.n_rejected) == (b.status, b.nfev, b.njev, b.nlu, b.n_accepted, b.n_rejected)
    assert np.array_equal(a.y_final, b.y_final)
    assert all(np.array_equal(x, y) for x, y in zip(a.t_events, b.t_events))
    y, st, *_ = oracle.radau(oracle.params_from_dict(p | {"dPhi_variable": 1}), N, y0, 0.0, 0.5, 1e-6, 1e-3, 1e-3)
    assert st.status == b.status == 0
    assert abs(b.nfev - st.nfev) <= max(6, 0.03 * st.nfev) and abs(b.n_accepted - st.n_accepted) <= 2
    assert np.max(np.abs(b.y_final - y)) < 2e-3
