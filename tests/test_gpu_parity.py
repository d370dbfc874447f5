"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle and the committed
golden vectors (which come from the reference's own code, oracle/make_goldens.py).

Tolerances (fp64).  The device evaluates the same formulas with fused multiply-adds, Newton
reciprocals instead of divisions, and pow(b, e) = exp(e log b); none of these is bit-identical to
numpy/libm, so results are compared relative to each field's max |value|:
  RHS                          <= 2e-13   (one evaluation; observed ~1e-15 .. 1e-14)
  RK4 / RK45 short runs        <= 1e-10   (tens to hundreds of steps; the system is dissipative)
  RK45 step sequence           identical accept/reject sequence, nfev equal
  end-to-end vs HDF5 goldens   the reference's own rtol 0.1 / atol 0.01 (tests/Regression_test/test_regression.py:29-30)
"""
import numpy as np
import pytest

from common import GOLDEN, noisy_state, parse_key, rel_to_max, scenario, synthetic_state

pytestmark = pytest.mark.gpu

RHS_TOL = 2e-13
RUN_TOL = 1e-10


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def make_model(p):
    from marlpde_amd.LHeureux_model import LMAHeureuxPorosityDiff
    return LMAHeureuxPorosityDiff.from_scenario(p, device=0)


def test_library_loaded_is_in_tree():
    from marlpde_amd import _abi
    lib = _abi.load()
    assert "integrating-diagenetic-equations-using-python_amd/csrc/libmarl_hip.so" in lib._name


def test_derived_constants_match_reference():
    import json
    gold = json.load(open(f"{GOLDEN}/derived_constants.json"))
    for name in ("default", "A", "matlab", "stiffphi"):
        eq = make_model(scenario(name))
        for k, v in gold[name].items():
            if k == "mask_cells":
                assert list(range(eq.mask_lo, eq.mask_hi)) == v
            else:
                assert getattr(eq, k) == pytest.approx(v, rel=1e-15, abs=0), (name, k)
        eq.close()


def test_rhs_and_events_against_reference_vectors():
    g = np.load(f"{GOLDEN}/rhs_vectors.npz")
    worst = 0.0
    for key in g["index"]:
        s, st, fv, N = parse_key(key)
        eq = make_model(scenario(s, N, fv))
        y = g[f"{key}|y"]
        r = eq.fun(0.0, y, None, None, None)
        assert r is not y and r.shape == y.shape
        err = rel_to_max(r, g[f"{key}|rate"])
        worst = max(worst, err)
        assert err <= RHS_TOL, (key, err)
        ev = eq.events_all(y)[0]
        ref = g[f"{key}|events"]
        assert np.max(np.abs(ev - ref) / np.maximum(1.0, np.abs(ref))) <= 1e-13, (key, ev, ref)
        # the scipy-style callables agree with the batched call
        assert eq.zeros_W(0.0, y) == ev[6] and eq.zeros(0.0, y) == ev[0]
        eq.close()
    print(f"worst RHS rel-to-max error over {len(g['index'])} reference vectors: {worst:.2e}")


@pytest.mark.parametrize("N", [200, 1000, 4096, 65536])
def test_rhs_device_layouts_against_oracle(torch_cuda, oracle, N):
    torch = torch_cuda
    from marlpde_amd._abi import LAYOUT_FIELD_MAJOR, LAYOUT_TILED
    p = scenario("A", N)
    eq = make_model(p)
    eq.use_stream(torch.cuda.current_stream().cuda_stream)
    y = noisy_state(p, N, seed=N)
    ref = oracle.rhs(oracle.params_from_model(eq), N, y)
    yd = torch.from_numpy(y).cuda()
    out = torch.empty_like(yd)
    eq.rhs_device(yd.data_ptr(), out.data_ptr(), LAYOUT_FIELD_MAJOR)
    torch.cuda.synchronize()
    assert rel_to_max(out.cpu().numpy(), ref) <= RHS_TOL
    # tiled layout: convert, evaluate, convert back
    nt = eq.state_doubles(LAYOUT_TILED)
    yt = torch.zeros(nt, dtype=torch.float64, device="cuda")
    rt = torch.zeros(nt, dtype=torch.float64, device="cuda")
    eq.convert_layout_device(yd.data_ptr(), yt.data_ptr(), LAYOUT_FIELD_MAJOR, LAYOUT_TILED)
    eq.rhs_device(yt.data_ptr(), rt.data_ptr(), LAYOUT_TILED)
    back = torch.empty_like(yd)
    eq.convert_layout_device(rt.data_ptr(), back.data_ptr(), LAYOUT_TILED, LAYOUT_FIELD_MAJOR)
    torch.cuda.synchronize()
    assert rel_to_max(back.cpu().numpy(), ref) <= RHS_TOL
    # round trip of the layout conversion is exact
    rt2 = torch.empty_like(yd)
    eq.convert_layout_device(yt.data_ptr(), rt2.data_ptr(), LAYOUT_TILED, LAYOUT_FIELD_MAJOR)
    torch.cuda.synchronize()
    assert torch.equal(rt2, yd)
    assert np.allclose(eq.events_device(yd.data_ptr())[0], oracle.events(oracle.params_from_model(eq), N, y), rtol=1e-13, atol=1e-13)
    eq.close()


RK4_VARIANT_DEPTH = (1, 2, 4, 8, 16)   # steps per launch of kRk4Variants (marl_api.hip)


@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("variant", range(5))
def test_rk4_fused_variants_against_oracle(torch_cuda, oracle, variant, layout):
    """Every shipped fused-RK4 instantiation (1 / 2 / 4 / 8 / 16 steps per launch, both layouts) on a grid that is not a
    multiple of any tile, incl. a step count that is not a multiple of the fused depth."""
    torch = torch_cuda
    from marlpde_amd._abi import LAYOUT_FIELD_MAJOR
    per = RK4_VARIANT_DEPTH[variant]
    N, nsteps = 5003, 2 * per + 3      # two launches of the variant's OWN depth + a remainder chain
    p = scenario("default", N)
    eq = make_model(p)
    eq.use_stream(torch.cuda.current_stream().cuda_stream)
    eq.set_option("rk4_variant", variant)
    y = synthetic_state(p, N, amplitude=0.05)
    dt = 0.25 * (eq.Depths.length / N) ** 2
    ref = oracle.rk4(oracle.params_from_model(eq), N, y, dt, nsteps)
    yd = torch.from_numpy(y).cuda()
    if layout == LAYOUT_FIELD_MAJOR:
        eq.integrate_rk4_device(yd.data_ptr(), dt, nsteps, layout)
        got = yd
    else:
        yt = torch.zeros(eq.state_doubles(layout), dtype=torch.float64, device="cuda")
        eq.convert_layout_device(yd.data_ptr(), yt.data_ptr(), 0, layout)
        eq.integrate_rk4_device(yt.data_ptr(), dt, nsteps, layout)
        got = torch.empty_like(yd)
        eq.convert_layout_device(yt.data_ptr(), got.data_ptr(), layout, 0)
    torch.cuda.synchronize()
    assert rel_to_max(got.cpu().numpy(), ref) <= RUN_TOL
    eq.close()


@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("N", [65536, 98304])
def test_rk4_default_kernel_at_config2_size(torch_cuda, oracle, N, layout):
    """BASELINE configs[1] (N = 65 536) and the largest grid that still takes the 16-steps-per-launch default: 37 steps =
    2 launches of rk4_fused_kernel<256,1,*,16> (transcendental reuse live, expansion centre carried over 16 steps)
    + 4 + 1, both device layouts, against the oracle; then the same with every evaluation forced onto its full path."""
    torch = torch_cuda
    p = scenario("default", N)
    eq = make_model(p)
    eq.use_stream(torch.cuda.current_stream().cuda_stream)
    y = synthetic_state(p, N, amplitude=0.01)
    dt = 0.25 * (eq.Depths.length / N) ** 2
    ref = oracle.rk4(oracle.params_from_model(eq), N, y, dt, 37, omp=True)
    for no_reuse in (0, 1):
        eq.set_option("no_reuse", no_reuse)
        yd = torch.from_numpy(y).cuda()
        buf = torch.zeros(eq.state_doubles(layout), dtype=torch.float64, device="cuda")
        eq.convert_layout_device(yd.data_ptr(), buf.data_ptr(), 0, layout)
        eq.integrate_rk4_device(buf.data_ptr(), dt, 37, layout)
        got = torch.empty_like(yd)
        eq.convert_layout_device(buf.data_ptr(), got.data_ptr(), layout, 0)
        torch.cuda.synchronize()
        assert rel_to_max(got.cpu().numpy(), ref) <= RUN_TOL, no_reuse
    eq.close()


@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("N,variant,nsteps", [(5003, 2, 4 * 7 + 3), (5003, 3, 8 * 3 + 1), (5003, 0, 9), (5003, 1, 2 * 6 + 1), (700, 2, 4 * 5),
                                              (65536, -1, 16 * 3 + 5), (1 << 20, -1, 4 * 10 + 2), (300001, -1, 4 * 9), (98304, -1, 16)])
def test_rk4_streamed_loop_is_bit_identical_to_per_level_launches(torch_cuda, N, variant, nsteps, layout):
    """The default fixed-step path of one large grid is ONE dataflow launch over (level, tile) work items
    (rk4_stream_kernel: tiles wait for their three producers of the previous level instead of for a launch boundary).  Every
    item computes what the workgroup of the per-level launch computes, so the result must be bit-identical to the launch-per-
    level path (option rk4_stream = 0) - for few tiles (3 at N = 700), many (4682 at N = 2^20), step counts that leave a
    remainder chain, several fused depths and both layouts (rk4_stream = 2 forces the streamed path below the size from which it
    is the default); marl_synchronize reports a streamed run that gave up waiting."""
    torch = torch_cuda
    p = scenario("default", N)
    eq = make_model(p)
    eq.use_stream(torch.cuda.current_stream().cuda_stream)
    if variant >= 0:
        eq.set_option("rk4_variant", variant)
    y = synthetic_state(p, N, amplitude=0.02)
    dt = 0.25 * (eq.Depths.length / N) ** 2
    out = []
    for stream in (0, 2, 2, 2, 2, 1):   # per-level launches, four forced streamed runs (a race would not show every time), the default
        eq.set_option("rk4_stream", stream)
        yd = torch.from_numpy(y).cuda()
        buf = torch.zeros(eq.state_doubles(layout), dtype=torch.float64, device="cuda")
        eq.convert_layout_device(yd.data_ptr(), buf.data_ptr(), 0, layout)
        eq.integrate_rk4_device(buf.data_ptr(), dt, nsteps, layout)
        eq.synchronize()
        got = torch.empty_like(yd)
        eq.convert_layout_device(buf.data_ptr(), got.data_ptr(), layout, 0)
        eq.synchronize()
        out.append(got.cpu().numpy())
    assert np.all(np.isfinite(out[0]))
    for o in out[1:]:
        assert np.array_equal(o, out[0])
    eq.close()


@pytest.mark.parametrize("N,variant,nsteps,max_items", [(5003, 2, 4 * 11 + 1, 100), (5003, 2, 4 * 12, 60), (150, 2, 4 * 6, 0), (230, 0, 7, 0)])
def test_rk4_streamed_loop_split_launches_and_single_tile(torch_cuda, N, variant, nsteps, max_items):
    """A call with more (level, tile) items than the 32-bit item counter may hand out in one launch is split into launches of an
    even number of levels (forced here through the test hook rk4_stream_max_items: 23 tiles, 4 / 2 levels per launch, 11 / 12
    levels in all); and grids of one or two tiles (no left / right producer to wait for) - all bit-identical to per-level launches."""
    torch = torch_cuda
    p = scenario("default", N)
    eq = make_model(p)
    eq.use_stream(torch.cuda.current_stream().cuda_stream)
    eq.set_option("rk4_variant", variant)
    y = synthetic_state(p, N, amplitude=0.02)
    dt = 0.25 * (eq.Depths.length / N) ** 2
    out = []
    for stream in (0, 2, 2):
        eq.set_option("rk4_stream", stream)
        eq.set_option("rk4_stream_max_items", max_items)
        yd = torch.from_numpy(y).cuda()
        eq.integrate_rk4_device(yd.data_ptr(), dt, nsteps, 0)
        eq.synchronize()
        out.append(yd.cpu().numpy())
    assert np.all(np.isfinite(out[0]))
    assert np.array_equal(out[1], out[0]) and np.array_equal(out[2], out[0])
    eq.close()


def test_rk4_streamed_loop_with_time_varying_porosity_diffusion(torch_cuda, oracle):
    """The dPhi_variable instantiation of the streamed loop (rk4_stream_kernel<..., VD = true>): against the oracle, and bit-identical
    to per-level launches."""
    torch = torch_cuda
    N, nsteps = 5003, 4 * 5 + 1
    p = scenario("default", N, dPhi_variable=True)
    eq = make_model(p)
    eq.use_stream(torch.cuda.current_stream().cuda_stream)
    y = synthetic_state(p, N, amplitude=0.02)
    dt = 0.2 * (eq.Depths.length / N) ** 2
    ref = oracle.rk4(oracle.params_from_model(eq), N, y, dt, nsteps)
    out = []
    for stream in (0, 2, 2):
        eq.set_option("rk4_stream", stream)
        yd = torch.from_numpy(y).cuda()
        eq.integrate_rk4_device(yd.data_ptr(), dt, nsteps, 0)
        eq.synchronize()
        out.append(yd.cpu().numpy())
    assert rel_to_max(out[0], ref) <= RUN_TOL
    assert np.array_equal(out[1], out[0]) and np.array_equal(out[2], out[0])
    eq.close()


def test_rk4_streamed_loop_reports_a_raised_flag_and_recovers(torch_cuda):
    """The streamed loop's safety net: a workgroup that gives up waiting raises a device flag, waiting workgroups leave, and the
    asynchronous entry point's error surfaces at marl_synchronize (error -2, "the state is invalid"); the context then resets its
    counters and the next run is bit-identical to per-level launches again.  (The flag is raised through the test hook.)"""
    torch = torch_cuda
    from marlpde_amd._abi import MarlError
    N, nsteps, layout = 300001, 24, 1
    p = scenario("default", N)
    eq = make_model(p)
    eq.use_stream(torch.cuda.current_stream().cuda_stream)
    y = synthetic_state(p, N, amplitude=0.02)
    dt = 0.25 * (eq.Depths.length / N) ** 2
    yd = torch.from_numpy(y).cuda()
    buf = torch.zeros(eq.state_doubles(layout), dtype=torch.float64, device="cuda")

    def run(stream):
        eq.set_option("rk4_stream", stream)
        eq.convert_layout_device(yd.data_ptr(), buf.data_ptr(), 0, layout)
        eq.integrate_rk4_device(buf.data_ptr(), dt, nsteps, layout)
        eq.synchronize()
        return buf.clone()
    ref = run(0)
    assert torch.equal(run(1), ref)
    eq.set_option("rk4_stream_test_raise", 1)
    with pytest.raises(MarlError, match="state is invalid"):
        run(1)
    for _ in range(3):
        assert torch.equal(run(1), ref)
    eq.close()


@pytest.mark.parametrize("name,N", [("default", 200), ("A", 200), ("matlab", 1024), ("stiffphi", 64), ("A", 3000)])
def test_rk4_host_entry_against_oracle(oracle, name, N):
    """marl_integrate_rk4 (host pointers): small grids take the one-workgroup on-chip path, larger ones the fused path."""
    p = scenario(name, N)
    eq = make_model(p)
    y = noisy_state(p, N, seed=1, sigma=0.02)
    dt = 0.2 * (eq.Depths.length / N) ** 2
    got = eq.integrate_rk4(y, dt, 25)
    ref = oracle.rk4(oracle.params_from_model(eq), N, y, dt, 25)
    assert rel_to_max(got, ref) <= RUN_TOL
    eq.close()


@pytest.mark.parametrize("traj", ["rk45_traj_A_N200", "rk45_traj_default_N200", "rk45_traj_A_N64"])
def test_rk45_reproduces_scipy_trajectory(traj):
    """Same accepted/rejected sequence as scipy's RK45 on the reference RHS (golden from oracle/make_goldens.py)."""
    g = np.load(f"{GOLDEN}/{traj}.npz")
    name, N = traj.split("_")[2], int(traj.split("N")[-1])
    eq = make_model(scenario(name, N))
    te = g["t_eval"] if g["t_eval"].size else None
    res = eq.integrate_rk45(g["y0"], tuple(g["t_span"]), float(g["first_step"]), float(g["rtol"]), float(g["atol"]), t_eval=te)
    assert res.status == 0 and res.nfev == int(g["nfev"])
    assert res.n_accepted == len(g["step_times"]) - 1
    assert rel_to_max(res.y_final, g["y_final"]) <= 1e-9
    if te is not None:
        assert np.array_equal(res.t, g["t_eval"])
        assert np.max(np.abs(res.y - g["y_eval"])) <= 1e-9
    assert [len(e) for e in res.t_events] == list(g["n_events"])
    eq.close()


# The two schedules of the adaptive loop on one grid (same attempt arithmetic; different order of the error sum, table-driven
# instead of libm pow in the controller):
#   stream    ONE launch for a whole batch of attempts, resident workgroups meeting at a barrier in memory (rk45_stream_kernel);
#             the default up to three rounds of resident workgroups (~750 000 cells on an MI355X)
#   launches  attempt + first-level reduction + control kernels per attempt (the round-1..3 loop; the default above that size)
RK45_PATHS = {"default": {}, "stream": {"rk45_stream": 2}, "launches": {"rk45_stream": 0}}


def set_rk45_path(eq, path):
    for k, v in RK45_PATHS[path].items():
        eq.set_option(k, v)


@pytest.mark.parametrize("path", ["default", "stream", "launches"])
@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("N", [5003, 777])
def test_rk45_fused_large_grid_against_oracle(torch_cuda, oracle, N, layout, path):
    torch = torch_cuda
    variant = 0
    p = scenario("A", N)
    eq = make_model(p)
    eq.use_stream(torch.cuda.current_stream().cuda_stream)
    eq.set_option("rk45_variant", variant)
    set_rk45_path(eq, path)
    y = synthetic_state(p, N, amplitude=0.05)
    dx2 = (eq.Depths.length / N) ** 2
    t1 = 40 * dx2
    yref, st, steps, _, _ = oracle.rk45(oracle.params_from_model(eq), N, y, 0.0, t1, 0.5 * dx2, 1e-5, 1e-7)
    yd = torch.from_numpy(y).cuda()
    if layout == 0:
        res = eq.integrate_rk45_device(yd.data_ptr(), (0.0, t1), 0.5 * dx2, 1e-5, 1e-7, layout)
        got = yd
    else:
        yt = torch.zeros(eq.state_doubles(layout), dtype=torch.float64, device="cuda")
        eq.convert_layout_device(yd.data_ptr(), yt.data_ptr(), 0, layout)
        res = eq.integrate_rk45_device(yt.data_ptr(), (0.0, t1), 0.5 * dx2, 1e-5, 1e-7, layout)
        got = torch.empty_like(yd)
        eq.convert_layout_device(yt.data_ptr(), got.data_ptr(), layout, 0)
    torch.cuda.synchronize()
    assert (res.status, res.n_accepted, res.n_rejected, res.nfev) == (st.status, st.n_accepted, st.n_rejected, st.nfev)
    assert res.t_reached == t1
    assert rel_to_max(got.cpu().numpy(), yref) <= RUN_TOL
    eq.close()


@pytest.mark.parametrize("path", ["stream", "launches"])
def test_rk45_host_entry_large_grid_t_eval_and_budget(oracle, path):
    """Large-grid host entry: interior t_eval samples by dense output; attempt budget stops with status 2."""
    N = 3000
    p = scenario("default", N)
    eq = make_model(p)
    set_rk45_path(eq, path)
    y = synthetic_state(p, N, amplitude=0.03)
    dx2 = (eq.Depths.length / N) ** 2
    t1 = 30 * dx2
    te = np.array([0.0, 0.31 * t1, 0.5 * t1, t1])
    P = oracle.params_from_model(eq)
    yref, st, _, ye, _ = oracle.rk45(P, N, y, 0.0, t1, 0.4 * dx2, 1e-4, 1e-6, t_eval=te)
    res = eq.integrate_rk45(y, (0.0, t1), 0.4 * dx2, 1e-4, 1e-6, t_eval=te)
    assert res.status == 0 and res.nfev == st.nfev and res.n_accepted == st.n_accepted
    assert np.array_equal(res.t, te)
    assert np.array_equal(res.y[:, 0], y)
    for j in range(len(te)):
        assert rel_to_max(res.y[:, j], ye[j]) <= RUN_TOL, j
    assert rel_to_max(res.y_final, yref) <= RUN_TOL
    res2 = eq.integrate_rk45(y, (0.0, t1), 0.4 * dx2, 1e-4, 1e-6, max_attempts=7, events=False)
    _, st2, _, _, _ = oracle.rk45(P, N, y, 0.0, t1, 0.4 * dx2, 1e-4, 1e-6, max_attempts=7)
    assert res2.status == 2 == st2.status and res2.t_reached == pytest.approx(st2.t, rel=1e-12)
    eq.close()


def test_rk45_event_roots_match_scipy_golden():
    """Root times of a monitor that fires, against scipy's own t_events on the REFERENCE RHS (golden
    rk45_event_default_N400.npz, oracle/make_goldens.py gen_rk45_event): what the reference prints and stores
    (marlpde/Evolve_scenario.py:118-145, 175-177)."""
    g = np.load(f"{GOLDEN}/rk45_event_default_N400.npz")
    eq = make_model(scenario("default", 400))
    res = eq.integrate_rk45(g["y0"], tuple(g["t_span"]), float(g["first_step"]), float(g["rtol"]), float(g["atol"]))
    assert res.status == 0 and res.nfev == int(g["nfev"]) and res.n_accepted == len(g["step_times"]) - 1
    assert [len(e) for e in res.t_events] == list(g["n_events"]) and sum(g["n_events"]) >= 1
    got = np.concatenate(res.t_events)
    assert np.allclose(got, g["t_events"], rtol=1e-9, atol=1e-12 * float(g["t_span"][1]))
    assert rel_to_max(res.y_final, g["y_final"]) <= 1e-9
    eq.close()


def test_rk45_event_root_matches_oracle(oracle):
    """A monitor that changes sign: a porosity dip makes min(U) negative at t0; the reaction term lifts the
    porosity and zeros_U crosses zero at t ~ 1.2e-4.  Root times come from dense output + Brent on both sides
    (scipy/integrate/_ivp/ivp.py:51-76, 673-694)."""
    N = 400
    p = scenario("default", N)
    eq = make_model(p)
    L = eq.Depths.length
    x = eq.Depths.axes_coords[0]
    y = eq.get_state(p["CAIni"], p["CCIni"], p["cCaIni"], p["cCO3Ini"], p["PhiIni"])
    y[4] = 0.8 - 0.04 * np.exp(-((x - 0.5 * L) / (0.08 * L)) ** 2)
    y = y.ravel()
    dx2 = (L / N) ** 2
    P = oracle.params_from_model(eq)
    t1 = 1000 * dx2
    yref, st, _, _, tev = oracle.rk45(P, N, y, 0.0, t1, 0.5 * dx2, 1e-5, 1e-7)
    res = eq.integrate_rk45(y, (0.0, t1), 0.5 * dx2, 1e-5, 1e-7)
    assert [len(e) for e in tev] == [0, 0, 0, 0, 0, 1, 0], "test state must trigger zeros_U exactly once"
    assert [len(e) for e in res.t_events] == [len(e) for e in tev]
    for a, b in zip(res.t_events, tev):
        assert np.allclose(a, b, rtol=1e-9, atol=1e-12 * t1)
    assert res.n_accepted == st.n_accepted and rel_to_max(res.y_final, yref) <= 1e-9
    eq.close()


@pytest.mark.parametrize("path", ["stream", "launches"])
def test_rk45_event_root_on_a_large_grid_every_schedule(oracle, path):
    """The porosity-dip state of test_rk45_event_root_matches_oracle on a grid that is NOT one workgroup (N = 3000): the loop pauses on
    the monitor's sign change (pause_on_event: the events decide the status, so both schedules' controllers take them first), the
    host locates the root by dense output + Brent.  Root time, decisions and end state against the oracle, for each schedule; a run
    with a small attempts-per-launch setting exercises the persistent loop's counters across launches."""
    N = 1500
    p = scenario("default", N)
    eq = make_model(p)
    set_rk45_path(eq, path)
    if path == "stream":
        eq.set_option("rk45_stream_attempts", 7)
    L = eq.Depths.length
    x = eq.Depths.axes_coords[0]
    y = eq.get_state(p["CAIni"], p["CCIni"], p["cCaIni"], p["cCO3Ini"], p["PhiIni"])
    y[4] = 0.8 - 0.04 * np.exp(-((x - 0.5 * L) / (0.08 * L)) ** 2)
    y = y.ravel()
    dx2 = (L / N) ** 2
    P = oracle.params_from_model(eq)
    t1 = 3000 * dx2
    yref, st, _, _, tev = oracle.rk45(P, N, y, 0.0, t1, 0.5 * dx2, 1e-5, 1e-7)
    res = eq.integrate_rk45(y, (0.0, t1), 0.5 * dx2, 1e-5, 1e-7)
    assert sum(len(e) for e in tev) >= 1, "test state must trigger a monitor"
    assert [len(e) for e in res.t_events] == [len(e) for e in tev]
    for a, b in zip(res.t_events, tev):
        assert np.allclose(a, b, rtol=1e-9, atol=1e-12 * t1)
    assert (res.status, res.n_accepted, res.n_rejected, res.nfev) == (st.status, st.n_accepted, st.n_rejected, st.nfev)
    assert rel_to_max(res.y_final, yref) <= 1e-9
    eq.close()


@pytest.mark.parametrize("path", ["stream", "launches"])
def test_rk45_large_grid_schedules_with_time_varying_porosity_diffusion(torch_cuda, oracle, path):
    """The dPhi_variable instantiations of both schedules (rk45_stream_kernel<..., VD = true>, rk45_attempt_kernel<..., VD = true>) on a grid
    that is not one workgroup, against the oracle's restatement of the variant (field-major layout, rejected attempts included)."""
    torch = torch_cuda
    N = 2500
    p = scenario("A", N) | {"dPhi_variable": True}
    eq = make_model(p)
    eq.use_stream(torch.cuda.current_stream().cuda_stream)
    set_rk45_path(eq, path)
    y = synthetic_state(p, N, amplitude=0.05)
    dx2 = (eq.Depths.length / N) ** 2
    t1 = 40 * dx2
    yref, st, *_ = oracle.rk45(oracle.params_from_model(eq), N, y, 0.0, t1, 0.5 * dx2, 1e-5, 1e-7)
    yd = torch.from_numpy(y).cuda()
    res = eq.integrate_rk45_device(yd.data_ptr(), (0.0, t1), 0.5 * dx2, 1e-5, 1e-7, 0)
    torch.cuda.synchronize()
    assert (res.status, res.n_accepted, res.n_rejected, res.nfev) == (st.status, st.n_accepted, st.n_rejected, st.nfev) and st.n_rejected > 0
    assert rel_to_max(yd.cpu().numpy(), yref) <= RUN_TOL
    eq.close()


# (a tile of the attempt kernel advances 244 cells; an MI355X holds 4 x 256 = 1 024 resident workgroups of the persistent kernel)
@pytest.mark.parametrize("N", [1 << 20, 244 * 1024, 244 * 1024 + 1, 244 * 2048, 244 * 1024 + 244 * 1023],
                         ids=["4297-tiles", "1024-tiles-no-remainder", "1025-tiles-remainder-of-one", "2048-tiles-no-remainder", "2047-tiles"])
def test_rk45_persistent_loop_at_full_size_against_the_launch_per_attempt_loop(torch_cuda, N):
    """N = 2^20 (4 297 tiles on 1 024 resident workgroups: four static rounds + a remainder round handed out by the atomic counter):
    the persistent loop against one launch per attempt - same decisions, states equal to rounding (the two differ in the order of the
    error sum and in the controller's pow); the persistent loop twice - bit-identical, whoever computed the remainder tiles.  The other
    sizes are the edges of the remainder logic: whole rounds only (no remainder counter in play), a remainder of ONE tile, one tile short of
    two whole rounds."""
    torch = torch_cuda
    p = scenario("default", N)
    y = synthetic_state(p, N, amplitude=0.01)
    dx2 = ((p["max_depth"] / p["Xstar"]) / N) ** 2
    out = {}
    for tag, opts in (("stream", {"rk45_stream": 2}), ("stream2", {"rk45_stream": 2, "rk45_stream_attempts": 5}), ("fused", {"rk45_stream": 0})):   # ("fused": the launch-per-attempt loop)
        eq = make_model(p)
        eq.use_stream(torch.cuda.current_stream().cuda_stream)
        for k, v in opts.items():
            eq.set_option(k, v)
        yd = torch.from_numpy(y).cuda()
        res = eq.integrate_rk45_device(yd.data_ptr(), (0.0, 1e9), 0.5 * dx2, 1e-3, 1e-3, max_attempts=24)
        out[tag] = (res, yd.cpu().numpy())
        eq.close()
    key = lambda r: (r.status, r.n_accepted, r.n_rejected, r.nfev)  # noqa: E731
    assert key(out["stream"][0]) == key(out["fused"][0]) == key(out["stream2"][0]) and out["stream"][0].n_accepted >= 15
    assert np.array_equal(out["stream"][1], out["stream2"][1])
    assert out["stream"][0].t_reached == out["stream2"][0].t_reached
    assert rel_to_max(out["stream"][1], out["fused"][1]) <= 1e-11


def test_sweep_rk4_and_rk45_against_oracle(torch_cuda, oracle):
    """Batched sweep: one workgroup per instance, per-instance parameters, dt and controller."""
    torch = torch_cuda
    from marlpde_amd.LHeureux_model import LMAHeureuxPorosityDiff
    N, B = 1024, 24
    base = scenario("default", N)
    rng = np.random.default_rng(3)
    inst = [{"Phi0": float(a), "PhiIni": float(b), "PhiNR": float(b), "k3": float(k), "k4": float(k)}
            for a, b, k in zip(rng.uniform(0.5, 0.8, B), rng.uniform(0.5, 0.8, B), 10 ** rng.uniform(-2, -1, B))]
    eq = LMAHeureuxPorosityDiff.from_scenario(base, device=0, instances=inst)
    eq.use_stream(torch.cuda.current_stream().cuda_stream)
    assert eq.n_instances == B
    y0 = np.stack([synthetic_state(base | i, N, amplitude=0.02) for i in inst])
    dx2 = (eq.Depths.length / N) ** 2
    dts = rng.uniform(0.15, 0.3, B) * dx2
    yd = torch.from_numpy(y0).cuda()
    eq.sweep_rk4_device(yd.data_ptr(), dts, 20)
    torch.cuda.synchronize()
    got = yd.cpu().numpy()
    for b in range(B):
        ref = oracle.rk4(oracle.params_from_model(eq, b), N, y0[b], dts[b], 20)
        assert rel_to_max(got[b], ref) <= RUN_TOL, b
    # batched RHS and monitors
    r = eq.fun(0.0, y0.ravel()).reshape(B, -1)
    ev = eq.events_all(y0.ravel())
    for b in (0, B // 2, B - 1):
        P = oracle.params_from_model(eq, b)
        assert rel_to_max(r[b], oracle.rhs(P, N, y0[b])) <= RHS_TOL
        assert np.allclose(ev[b], oracle.events(P, N, y0[b]), rtol=1e-13, atol=1e-13)
    # adaptive sweep with an attempt budget (the bench's configuration) and to a common end time
    yd = torch.from_numpy(y0).cuda()
    res = eq.sweep_rk45_device(yd.data_ptr(), (0.0, 1.0), 0.5 * dx2, 1e-3, 1e-3, max_attempts=60)
    torch.cuda.synchronize()
    got = yd.cpu().numpy()
    for b in range(B):
        yref, st, _, _, _ = oracle.rk45(oracle.params_from_model(eq, b), N, y0[b], 0.0, 1.0, 0.5 * dx2, 1e-3, 1e-3, max_attempts=60)
        assert (res[b].status, res[b].n_accepted, res[b].n_rejected) == (2, st.n_accepted, st.n_rejected), b
        assert res[b].t_reached == pytest.approx(st.t, rel=1e-12)
        assert rel_to_max(got[b], yref) <= 1e-9, b
    eq.close()


def test_nan_state_is_data_not_error(oracle):
    """Phi <= 0 gives non-finite rates (log of a negative number) exactly where the reference's numba path does;
    a NaN error norm is a rejected step with factor 0.2 (scipy/integrate/_ivp/rk.py:163-164)."""
    N = 256
    p = scenario("default", N)
    eq = make_model(p)
    y = synthetic_state(p, N)
    y[4 * N + 17] = -0.1
    r = eq.fun(0.0, y)
    ref = oracle.rhs(oracle.params_from_model(eq), N, y)
    # every entry that is non-finite in the reference is non-finite here; the cell with Phi < 0 may carry MORE
    # non-finite rates (one shared reciprocal serves 1/Phi, 1/(1-Phi), 1/den, so its NaN reaches U as well) -
    # either way the error norm of a step through such a state is NaN and the step is rejected
    bad_ref, bad = ~np.isfinite(ref), ~np.isfinite(r)
    assert bad_ref.any() and np.all(bad[bad_ref])
    cells = np.unique(np.nonzero(bad)[0] % N)
    assert list(cells) == [17]
    ok = ~bad
    assert rel_to_max(np.where(ok, r, 0.0), np.where(ok, ref, 0.0)) <= RHS_TOL
    res = eq.integrate_rk45(y, (0.0, 1e-3), 1e-6, 1e-3, 1e-3, events=False)
    _, st, _, _, _ = oracle.rk45(oracle.params_from_model(eq), N, y, 0.0, 1e-3, 1e-6, 1e-3, 1e-3)
    assert res.status == st.status == -1 and res.n_accepted == st.n_accepted == 0
    eq.close()


@pytest.mark.parametrize("path", ["stream", "launches"])
def test_nan_state_on_a_large_grid_rejects_down_to_the_minimum_step(oracle, path):
    """The same invalid state (one cell with Phi < 0) on a grid that is not one workgroup, through both schedules of the adaptive loop:
    every attempt's error norm is NaN - a NaN sum through the barrier's fixed-order addition / the record reduction, extrema that are never
    read - so every attempt is rejected with factor 0.2 until the step falls below 10 ulp(t): status -1 after exactly the oracle's
    number of rejections, nothing accepted."""
    N = 3000
    p = scenario("default", N)
    eq = make_model(p)
    set_rk45_path(eq, path)
    y = synthetic_state(p, N)
    y[4 * N + 1717] = -0.1
    res = eq.integrate_rk45(y, (0.0, 1e-3), 1e-6, 1e-3, 1e-3, events=False)
    _, st, *_ = oracle.rk45(oracle.params_from_model(eq), N, y, 0.0, 1e-3, 1e-6, 1e-3, 1e-3)
    assert res.status == st.status == -1 and res.n_accepted == st.n_accepted == 0 and res.n_rejected == st.n_rejected > 10
    eq.close()


def test_full_size_rk4_against_oracle_and_layout_agreement(torch_cuda, oracle):
    """BASELINE headline size N = 2^20: a few fused steps against the oracle, and the two device layouts /
    two kernel variants against each other."""
    torch = torch_cuda
    N, nsteps = 1 << 20, 8
    p = scenario("default", N)
    eq = make_model(p)
    eq.use_stream(torch.cuda.current_stream().cuda_stream)
    y = synthetic_state(p, N)
    dt = 0.25 * (eq.Depths.length / N) ** 2
    ref = oracle.rk4(oracle.params_from_model(eq), N, y, dt, nsteps, omp=True)
    results = []
    for variant, layout in ((2, 1), (2, 0), (0, 1), (3, 1)):   # 4 steps per launch (the default here), 1 and 8
        eq.set_option("rk4_variant", variant)
        yd = torch.from_numpy(y).cuda()
        buf = yd
        if layout:
            buf = torch.zeros(eq.state_doubles(layout), dtype=torch.float64, device="cuda")
            eq.convert_layout_device(yd.data_ptr(), buf.data_ptr(), 0, layout)
        eq.integrate_rk4_device(buf.data_ptr(), dt, nsteps, layout)
        if layout:
            eq.convert_layout_device(buf.data_ptr(), yd.data_ptr(), layout, 0)
        torch.cuda.synchronize()
        results.append(yd.cpu().numpy())
        assert rel_to_max(results[-1], ref) <= RUN_TOL, (variant, layout)
    for other in results[1:]:
        assert rel_to_max(other, results[0]) <= 1e-13
    eq.close()


def test_integrate_equations_rk45_against_reference_golden():
    """End-to-end drop-in: Scenario A to T* with RK45 on the GPU vs the reference's HDF5 golden, at the
    reference's own tolerance (tests/Regression_test/test_regression.py:21-53)."""
    from dataclasses import asdict, replace
    from marlpde_amd.Evolve_scenario import integrate_equations
    from marlpde_amd.parameters import Solver, Tracker
    gold = np.load(f"{GOLDEN}/ref_final_scenarioA_Phi0_0.6_PhiIni_0.5.npy")
    last, covered, depths, Xstar, folder = integrate_equations(
        asdict(replace(Solver(), method="RK45")), asdict(Tracker()), scenario("A"), results_root=None, verbose=False)
    assert last.shape == (5, 200) and covered == pytest.approx(13190.0) and folder is None and Xstar == 1319.0
    np.testing.assert_allclose(last, gold, rtol=0.1, atol=0.01)
    print("max abs deviation from the reference golden per field:", np.max(np.abs(last - gold), axis=1))


def test_interior_frames_against_the_reference_stored_frames(tmp_path):
    """The reference's HDF5 golden stores 101 frames; its tests read only the last (tests/Regression_test/
    test_regression.py:26-27).  Scenario A with t_eval = linspace(0, 1, 101) (marlpde/parameters.py:252-261): frames 10, 25, 50
    (t = 0.1, 0.25, 0.5 T*) at the reference's own tolerance, read back from the result file like a user would."""
    from dataclasses import asdict, replace
    from marlpde_amd.Evolve_scenario import integrate_equations
    from marlpde_amd.parameters import Solver, Tracker
    frames = np.load(f"{GOLDEN}/ref_frames_scenarioA_t0.1_0.25_0.5.npy")   # (3, 5, 200)
    tracker = asdict(Tracker()) | {"t_eval": np.linspace(0.0, 1.0, 101), "no_t_eval": 101}
    last, covered, _, _, folder = integrate_equations(asdict(replace(Solver(), method="RK45")), tracker, scenario("A"),
                                                       results_root=str(tmp_path) + "/", verbose=False)
    stored = np.load(folder + "LMAHeureuxPorosityDiff.npz")
    assert stored["solutions"].shape == (5, 200, 101) and np.allclose(stored["times"], np.linspace(0, 1, 101))
    for k, j in enumerate((10, 25, 50)):
        np.testing.assert_allclose(stored["solutions"][:, :, j], frames[k], rtol=0.1, atol=0.01)
    assert np.array_equal(stored["solutions"][:, :, -1], last)


def _run_slabs(torch, p, N, P, y0, t1, h0, rtol, atol, max_attempts=0):
    """Three slab contexts on ONE GPU stand for three ranks; strips and records are moved by plain tensor copies
    where the multi-GPU driver (marlpde_amd/domain.py) uses RCCL send/recv and all-gather."""
    from marlpde_amd.domain import HALO, STRIP, HipSlabEngine, owned_slice, partition
    parts = partition(N, P)
    eng = [HipSlabEngine(p, N, b, e, 0) for b, e in parts]
    ys = [torch.from_numpy(owned_slice(y0, N, b, e)).cuda() for b, e in parts]
    n = STRIP * HALO
    z = lambda k=n: torch.zeros(k, dtype=torch.float64, device="cuda")  # noqa: E731
    slo, shi, rlo, rhi = ([z() for _ in range(P)] for _ in range(4))
    rec, recs = [z(8) for _ in range(P)], z(8 * P)

    def halo_round(which):
        for r in range(P):
            eng[r].pack(which, slo[r], shi[r])
        for r in range(P):
            if r > 0:
                rlo[r].copy_(shi[r - 1])
            if r < P - 1:
                rhi[r].copy_(slo[r + 1])
        for r in range(P):
            eng[r].unpack(which, rlo[r], rhi[r])

    def gather():
        recs.copy_(torch.cat(rec))

    for r in range(P):
        eng[r].load(ys[r])
    halo_round(0)
    for r in range(P):
        eng[r].rhs0()
    halo_round(0)
    for r in range(P):
        eng[r].monitors(rec[r])
    gather()
    for r in range(P):
        eng[r].init_control(recs, P, 0.0, t1, h0, rtol, atol, max_attempts)
    for _ in range(200):
        for _ in range(8):
            for r in range(P):
                eng[r].attempt(rec[r])
            halo_round(-1)
            gather()
            for r in range(P):
                eng[r].control(recs, P)
        stats = [eng[r].status() for r in range(P)]
        if stats[0].status != 1:
            break
    for r in range(P):
        eng[r].store(ys[r])
    torch.cuda.synchronize()
    got = np.concatenate([y.cpu().numpy().reshape(5, -1) for y in ys], axis=1)
    for e in eng:
        e.close()
    return stats, got


@pytest.mark.parametrize("vd", [False, True])
def test_domain_decomposition_slabs_on_one_gpu(torch_cuda, oracle, vd):
    """BASELINE config 5 logic on ONE GPU: checks the slab kernels (pack / unpack / halo-consuming fused attempt /
    shared control) against the oracle's single-grid run, incl. rejected attempts (and with the dPhi_variable kernels)."""
    N, P = 5000, 3
    p = scenario("A", N) | {"dPhi_variable": vd}
    y0 = synthetic_state(p, N, amplitude=0.05)
    dx2 = ((p["max_depth"] / p["Xstar"]) / N) ** 2
    t1, h0, rtol, atol = 40 * dx2, 0.5 * dx2, 1e-5, 1e-7
    yref, st, *_ = oracle.rk45(oracle.params_from_dict(p), N, y0, 0.0, t1, h0, rtol, atol)
    stats, got = _run_slabs(torch_cuda, p, N, P, y0, t1, h0, rtol, atol)
    assert {(s.status, s.n_accepted, s.n_rejected, s.nfev, s.t) for s in stats} == {(0, st.n_accepted, st.n_rejected, st.nfev, t1)}
    assert rel_to_max(got, yref.reshape(5, N)) <= RUN_TOL


@pytest.mark.parametrize("transport", ["native", "rccl1", "rccl1-stream", "torch"])
def test_domain_driver_one_slab_equals_single_grid(torch_cuda, monkeypatch, transport):
    """DomainDecomposedRK45 as ONE slab on one GPU through each transport - the library loop without a communicator, the
    library loop with a one-rank RCCL communicator (ncclCommInitRank / ncclAllGather really run), and the host loop - takes
    the decisions of the single-grid integrator and reaches its state."""
    torch = torch_cuda
    from marlpde_amd.domain import DomainDecomposedRK45
    dd_stream = transport == "rccl1-stream"   # (option dd_stream: the slab's attempt as one launch of the persistent kernel; opt-in)
    transport = transport.split("-")[0]
    monkeypatch.setenv("MARL_DD_TRANSPORT", transport)
    if dd_stream:
        monkeypatch.setenv("MARL_HIP_OPTIONS", "dd_stream=2")
    N = 40000
    p = scenario("default", N)
    y = synthetic_state(p, N, amplitude=0.03)
    dx2 = ((p["max_depth"] / p["Xstar"]) / N) ** 2
    eq = make_model(p)
    eq.use_stream(torch.cuda.current_stream().cuda_stream)
    # the single-grid schedule with the transport's arithmetic: one slab without a communicator runs the single-grid integrator itself
    # (the persistent loop at this size: same bits); the library loop and the host loop reduce through the per-workgroup records like
    # the launch-per-attempt loop (same bits); with dd_stream the slab's attempt is one launch of the persistent kernel whose last
    # workgroup packs the message - the persistent loop's error sum, but the common decision is then taken by
    # slab_unpack_control_kernel with libm's pow where the persistent loop's controller uses the LDS tables (step sizes equal to a few
    # ulp, states to rounding)
    set_rk45_path(eq, "stream" if dd_stream else {"native": "default", "rccl1": "launches", "torch": "launches"}[transport])
    yd = torch.from_numpy(y).cuda()
    ref = eq.integrate_rk45_device(yd.data_ptr(), (0.0, 60 * dx2), 0.5 * dx2, 1e-5, 1e-7)
    dd = DomainDecomposedRK45(p, N, device=0)
    assert ("library loop" in dd.transport) == (transport != "torch"), dd.transport
    assert ("ncclAllGather" in dd.transport) == (transport == "rccl1"), dd.transport
    own = torch.from_numpy(y.copy()).cuda()
    st = dd.integrate(own, (0.0, 60 * dx2), 0.5 * dx2, 1e-5, 1e-7)
    assert (st.status, st.n_accepted, st.n_rejected, st.nfev) == (ref.status, ref.n_accepted, ref.n_rejected, ref.nfev)
    assert rel_to_max(own.cpu().numpy(), yd.cpu().numpy()) <= (1e-11 if dd_stream else 1e-13)
    dd.close()
    eq.close()


@pytest.mark.parametrize("P", [3, 8])
def test_config5_full_size_decomposition_equals_single_grid(torch_cuda, P):
    """BASELINE config 5 at its full size (N = 2^22, RK45): the decomposition is invisible - P slabs (8 = the north star's rank
    count; 3 = uneven sizes) with exchanged halos take the same accept/reject decisions as the single-grid integrator (itself
    checked against the oracle at the sizes the oracle finishes) and reach the same state, for an attempt budget of 12."""
    torch = torch_cuda
    N, budget = 1 << 22, 12
    p = scenario("default", N)
    y0 = synthetic_state(p, N, amplitude=0.01)
    dx2 = ((p["max_depth"] / p["Xstar"]) / N) ** 2
    t1, h0, rtol, atol = 1.0, 0.5 * dx2, 1e-3, 1e-3
    eq = make_model(p)
    eq.use_stream(torch.cuda.current_stream().cuda_stream)
    yd = torch.from_numpy(y0).cuda()
    res = eq.integrate_rk45_device(yd.data_ptr(), (0.0, t1), h0, rtol, atol, max_attempts=budget)
    single = yd.cpu().numpy().reshape(5, N)
    eq.close()
    stats, got = _run_slabs(torch, p, N, P, y0, t1, h0, rtol, atol, max_attempts=budget)
    assert res.status == 2 and res.n_accepted + res.n_rejected == budget and res.n_accepted >= 3
    assert {(s.status, s.n_accepted, s.n_rejected) for s in stats} == {(2, res.n_accepted, res.n_rejected)}
    assert all(s.t == pytest.approx(res.t_reached, rel=1e-12) for s in stats)   # (error-norm partial sums are grouped differently)
    assert np.max(np.abs(got - single)) <= 1e-13


@pytest.mark.parametrize("name,gold_file,first_step", [("matlab", "ref_matlab_Phi_0.5_k3_k4_0.01.npy", 1e-6),
                                                        ("default", "ref_final_high_porosity_0.8.npy", 5e-7)])
def test_integrate_equations_other_reference_cases(name, gold_file, first_step):
    """The reference's other two regression cases end to end with RK45 on the GPU, at the reference's tolerances
    (tests/Regression_test/test_regression.py:55-88 high porosity; :90-148 Matlab cross-check)."""
    from dataclasses import asdict, replace
    from marlpde_amd.Evolve_scenario import integrate_equations
    from marlpde_amd.parameters import Solver, Tracker
    gold = np.load(f"{GOLDEN}/{gold_file}")
    last, covered, *_ = integrate_equations(asdict(replace(Solver(), method="RK45", first_step=first_step)), asdict(Tracker()),
                                            scenario(name), results_root=None, verbose=False)
    assert covered == pytest.approx(13190.0)
    if name == "matlab":
        xs = (np.arange(200) + 0.5) * 2.5
        interp = np.stack([np.interp(xs, np.linspace(0, 500, 201), gold[f]) for f in range(5)])
        np.testing.assert_allclose(last[:, 2:], interp[:, 2:], atol=0.05)
    else:
        np.testing.assert_allclose(last, gold, rtol=0.1, atol=0.01)


def test_scipy_driven_radau_with_hip_rhs():
    """The reference's DEFAULT path: scipy's implicit Radau drives, the RHS (and the seven monitors) run on the GPU
    through the solve_ivp callable surface fun(t, y, *args) - Scenario A against the reference's golden."""
    from dataclasses import asdict
    from marlpde_amd.Evolve_scenario import integrate_equations
    from marlpde_amd.parameters import Solver, Tracker
    gold = np.load(f"{GOLDEN}/ref_final_scenarioA_Phi0_0.6_PhiIni_0.5.npy")
    last, covered, *_ = integrate_equations(asdict(Solver()) | {"scipy_driver": True}, asdict(Tracker()), scenario("A"), results_root=None,
                                            verbose=False)
    assert covered == pytest.approx(13190.0)
    np.testing.assert_allclose(last, gold, rtol=0.1, atol=0.01)
    assert np.max(np.abs(last - gold)) < 1e-3      # the stub-hosted reference itself is within 7e-5 of this golden


@pytest.mark.parametrize("method", ["RK23", "DOP853", "LSODA"])
def test_every_other_method_of_the_reference_solver_runs_through_scipy_on_the_hip_rhs(oracle, method):
    """The reference's Solver accepts ANY solve_ivp method (marlpde/parameters.py:205-219: LSODA with lband = uband = 1, an explicit
    method via replace(Solver(), method=...)); here every method without a native time loop is driven by scipy on the HIP RHS and
    the HIP monitors (Evolve_scenario.py, the solve_ivp branch, the reference's :100-109).  Checked against the SAME scipy call on
    the oracle's RHS: statistics, the final state and the result tuple.  Short span: explicit methods are stability-limited."""
    from dataclasses import asdict, replace
    from scipy.integrate import solve_ivp
    from marlpde_amd.Evolve_scenario import integrate_equations
    from marlpde_amd.parameters import Solver, Tracker
    N, t1 = 200, 2e-4
    p = scenario("A", N)
    solver = asdict(replace(Solver(), method=method, t_span=(0.0, t1)))
    tracker = asdict(Tracker()) | {"t_eval": np.linspace(0.0, t1, 2)}
    last, covered, depths, Xstar, folder = integrate_equations(solver, tracker, p, results_root=None, verbose=False)
    assert covered == pytest.approx(p["Tstar"] * t1) and folder is None and Xstar == p["Xstar"] and depths.N == N
    assert last.shape == (5, N) and np.all(np.isfinite(last))

    P = oracle.params_from_dict(p)
    y0 = np.stack([np.full(N, p[k]) for k in ("CAIni", "CCIni", "cCaIni", "cCO3Ini", "PhiIni")]).ravel()
    extra = {"lband": 1, "uband": 1} if method == "LSODA" else {}
    ref = solve_ivp(lambda t, y: oracle.rhs(P, N, y), (0.0, t1), y0, method=method, t_eval=tracker["t_eval"], first_step=1e-6,
                    rtol=1e-3, atol=1e-3, events=[lambda t, y, e=e: oracle.events(P, N, y)[e] for e in range(7)], **extra)
    assert ref.status == 0
    # explicit methods: the two RHS differ by ~1e-15 relative and the controller saw-tooths at the stability limit (SURVEY 6:
    # fun vs fun_numba under RK45 end 1.4e-10 apart); LSODA's Newton / order decisions amplify the same noise up to its tolerance
    tol = 2e-3 if method == "LSODA" else 1e-7
    assert np.max(np.abs(last - ref.y[:, -1].reshape(5, N))) <= tol, method


# ---------------------------------------------------------------------------------------------------------
# edge cases the reference's formulas cover but its own tests do not
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("overrides", [{"m1": 0.0}, {"n2": 0.0, "m2": 0.0}, {"FV_switch": 0}])
def test_rhs_unusual_parameters_against_oracle(oracle, overrides):
    """Exponent 0 makes pow(0, e) = 1 instead of 0 (the kernels' general `generic_p0` combination); FV off."""
    N = 96
    p = scenario("default", N) | overrides
    eq = make_model(p)
    P = oracle.params_from_model(eq)
    x = eq.Depths.axes_coords[0]
    for kind in ("under", "over"):
        y = eq.get_state(p["CAIni"], p["CCIni"], p["cCaIni"], p["cCO3Ini"], p["PhiIni"])
        if kind == "over":
            y[2] = 1.5 * (1 + 0.2 * np.sin(40 * x))
            y[3] = 1.4 * (1 + 0.2 * np.cos(31 * x))
        y[4] *= 1 + 0.1 * np.sin(25 * x)
        y = y.ravel()
        assert rel_to_max(eq.fun(0.0, y), oracle.rhs(P, N, y)) <= RHS_TOL, (overrides, kind)
    dt = 0.05 * (eq.Depths.length / N) ** 2
    assert rel_to_max(eq.integrate_rk4(y, dt, 10), oracle.rk4(P, N, y, dt, 10)) <= RUN_TOL
    eq.close()


@pytest.mark.parametrize("N,instances", [(200, 5), (300, 3), (513, 2), (64, 7)])
def test_sweep_window_selection_for_odd_sizes(torch_cuda, oracle, N, instances):
    """One-workgroup sweeps pick the smallest window >= N (256-, 512-, 1024-cell variants); cells beyond N idle."""
    torch = torch_cuda
    from marlpde_amd.LHeureux_model import LMAHeureuxPorosityDiff
    base = scenario("A", N)
    inst = [{"Phi0": 0.55 + 0.04 * i, "PhiIni": 0.5 + 0.03 * i, "PhiNR": 0.5 + 0.03 * i} for i in range(instances)]
    eq = LMAHeureuxPorosityDiff.from_scenario(base, device=0, instances=inst)
    y0 = np.stack([synthetic_state(base | i, N, amplitude=0.03) for i in inst])
    dx2 = (eq.Depths.length / N) ** 2
    yd = torch.from_numpy(y0).cuda()
    res = eq.sweep_rk45_device(yd.data_ptr(), (0.0, 60 * dx2), 0.3 * dx2, 1e-4, 1e-6)
    got = yd.cpu().numpy()
    for b in range(instances):
        yref, st, *_ = oracle.rk45(oracle.params_from_model(eq, b), N, y0[b], 0.0, 60 * dx2, 0.3 * dx2, 1e-4, 1e-6)
        assert (res[b].status, res[b].n_accepted, res[b].n_rejected) == (0, st.n_accepted, st.n_rejected)
        assert rel_to_max(got[b], yref) <= 1e-9
    eq.close()


@pytest.mark.parametrize("N", [12000, 40000])
def test_stage_reuse_boundary_regime_against_oracle(torch_cuda, oracle, N):
    """Grid sizes where the stage-to-stage expansions (marl_math.h) are partly in and partly out of range:
    some waves expand, others fall back to the full evaluation - the result must not care."""
    torch = torch_cuda
    p = scenario("default", N)
    eq = make_model(p)
    eq.use_stream(torch.cuda.current_stream().cuda_stream)
    y = synthetic_state(p, N, amplitude=0.05)
    dx2 = (eq.Depths.length / N) ** 2
    ref = oracle.rk4(oracle.params_from_model(eq), N, y, 0.25 * dx2, 12, omp=True)
    yd = torch.from_numpy(y).cuda()
    eq.integrate_rk4_device(yd.data_ptr(), 0.25 * dx2, 12)
    torch.cuda.synchronize()
    assert rel_to_max(yd.cpu().numpy(), ref) <= RUN_TOL
    yref, st, *_ = oracle.rk45(oracle.params_from_model(eq), N, y, 0.0, 30 * dx2, 0.5 * dx2, 1e-5, 1e-7, omp=True)
    yd = torch.from_numpy(y).cuda()
    res = eq.integrate_rk45_device(yd.data_ptr(), (0.0, 30 * dx2), 0.5 * dx2, 1e-5, 1e-7)
    assert (res.status, res.n_accepted, res.n_rejected) == (0, st.n_accepted, st.n_rejected)
    assert rel_to_max(yd.cpu().numpy(), yref) <= RUN_TOL
    eq.close()


def test_api_errors_are_reported_not_swallowed():
    from marlpde_amd._abi import MarlError
    p = scenario("default", 200)
    eq = make_model(p)
    y = synthetic_state(p, 200)
    with pytest.raises(MarlError, match="unknown option"):
        eq.set_option("no_such_knob", 1)
    with pytest.raises(MarlError, match="first_step"):
        eq.integrate_rk45(y, (0.0, 1e-3), 1.0, 1e-3, 1e-3)          # scipy: "`first_step` exceeds bounds"
    with pytest.raises(MarlError, match="t_eval"):
        eq.integrate_rk45(y, (0.0, 1e-3), 1e-6, 1e-3, 1e-3, t_eval=[0.0, 2e-3])
    with pytest.raises(ValueError, match="state has"):
        eq.fun(0.0, y[:-1])
    eq.close()


def test_integrate_equations_stores_results_like_the_reference(tmp_path):
    """Dataset names of the result file: solutions (5, N, n_t), times, event_0..6 (marlpde/Evolve_scenario.py:172-178)."""
    from dataclasses import asdict, replace
    from marlpde_amd.Evolve_scenario import integrate_equations
    from marlpde_amd.parameters import Solver, Tracker
    sol = asdict(replace(Solver(), method="RK45", t_span=(0, 2e-3)))
    trk = asdict(Tracker()) | {"t_eval": np.linspace(0, 2e-3, 3)}
    last, covered, depths, Xstar, folder = integrate_equations(sol, trk, scenario("A"), results_root=str(tmp_path) + "/", verbose=False)
    z = np.load(folder + "LMAHeureuxPorosityDiff.npz")
    assert z["solutions"].shape == (5, 200, 3) and np.array_equal(z["times"], trk["t_eval"])
    assert all(f"event_{i}" in z for i in range(7)) and float(z["attr_Phi0"]) == 0.6
    assert np.array_equal(z["solutions"][:, :, -1], last) and covered == pytest.approx(13190.0 * 2e-3)


@pytest.mark.parametrize("name,N,fv,dtf", [("A", 200, 1, 0.2), ("default", 64, 1, 0.2), ("matlab", 1000, 0, 0.2), ("default", 8, 1, 0.01),
                                            ("A", 3000, 1, 0.2)])   # 3000: the multi-workgroup fused kernels
def test_time_varying_porosity_diffusion_against_oracle(torch_cuda, oracle, name, N, fv, dtf):
    """SURVEY 8(f) rank 4 (`dPhi_variable`): RHS, fused RK4 and RK45 with dPhi = auxcon F Phi^3/(1-Phi).  The porosity
    Peclet number is 0.9 (N = 200) ... 22 (N = 8) here: the Fiadeiro-Veronis weight of Phi is active and uses the
    cell's own dPhi.  Parity is against the oracle only (the reference keeps this variant commented out)."""
    torch = torch_cuda
    p = scenario(name, N, fv=fv) | {"dPhi_variable": True}
    eq = make_model(p)
    eq.use_stream(torch.cuda.current_stream().cuda_stream)
    P = oracle.params_from_model(eq)
    assert P.dPhi_variable == 1
    y = noisy_state(p, N, seed=5, sigma=0.03)
    ref = oracle.rhs(P, N, y)
    assert rel_to_max(eq.fun(0.0, y), ref) <= RHS_TOL
    plain = make_model(scenario(name, N, fv=fv))
    assert rel_to_max(plain.fun(0.0, y), ref) > 1e-6            # not the fixed-coefficient result
    plain.close()
    y = synthetic_state(p, N, amplitude=0.05)
    dx2 = (eq.Depths.length / N) ** 2
    ref4 = oracle.rk4(P, N, y, dtf * dx2, 25)
    assert np.all(np.isfinite(ref4))
    assert rel_to_max(eq.integrate_rk4(y, dtf * dx2, 25), ref4) <= RUN_TOL
    yref, st, *_ = oracle.rk45(P, N, y, 0.0, 40 * dx2, 0.5 * dx2, 1e-5, 1e-7)
    yd = torch.from_numpy(y).cuda()
    res = eq.integrate_rk45_device(yd.data_ptr(), (0.0, 40 * dx2), 0.5 * dx2, 1e-5, 1e-7)
    assert (res.status, res.n_accepted, res.n_rejected) == (0, st.n_accepted, st.n_rejected)
    assert rel_to_max(yd.cpu().numpy(), yref) <= RUN_TOL
    eq.close()


@pytest.mark.parametrize("N", [3000, 700])
def test_mixed_upwind_direction_inside_waves(torch_cuda, oracle, N):
    """The fused kernels skip the upwind selects (and the right-hand LDS reads of the solids) when every lane of a
    wave has U > 0.  A 15 % porosity modulation makes U change sign every ~40 cells in the default scenario
    (presum = -3.2): waves mix both directions and must take the per-lane selects."""
    torch = torch_cuda
    p = scenario("default", N)
    eq = make_model(p)
    eq.use_stream(torch.cuda.current_stream().cuda_stream)
    P = oracle.params_from_model(eq)
    y = synthetic_state(p, N, amplitude=0.15, waves=N // 75)
    Phi = y.reshape(5, N)[4]
    U = eq.presum + eq.rhorat * Phi ** 3 * (1 - np.exp(10 - 10 / Phi)) / (1 - Phi)
    assert 0.2 < np.mean(U > 0) < 0.8
    assert rel_to_max(eq.fun(0.0, y), oracle.rhs(P, N, y)) <= RHS_TOL
    dx2 = (eq.Depths.length / N) ** 2
    ref = oracle.rk4(P, N, y, 0.1 * dx2, 12)
    yd = torch.from_numpy(y).cuda()
    eq.integrate_rk4_device(yd.data_ptr(), 0.1 * dx2, 12)
    assert rel_to_max(yd.cpu().numpy(), ref) <= RUN_TOL
    yref, st, *_ = oracle.rk45(P, N, y, 0.0, 20 * dx2, 0.2 * dx2, 1e-5, 1e-7)
    yd = torch.from_numpy(y).cuda()
    res = eq.integrate_rk45_device(yd.data_ptr(), (0.0, 20 * dx2), 0.2 * dx2, 1e-5, 1e-7)
    assert (res.status, res.n_accepted, res.n_rejected) == (0, st.n_accepted, st.n_rejected)
    assert rel_to_max(yd.cpu().numpy(), yref) <= RUN_TOL
    eq.close()


def test_config3_full_size_sweep_properties(torch_cuda, oracle):
    """BASELINE config 3 at its full size (4096 instances x N = 1024, RK45): size-independent properties instead of
    4096 oracle runs - instances with equal parameters end bit-identical wherever they sit in the batch (the
    16 x 16 x 16 grid is laid out twice over 8 distinct parameter sets), and sampled instances equal the oracle."""
    torch = torch_cuda
    from marlpde_amd.LHeureux_model import LMAHeureuxPorosityDiff
    N, B, D = 1024, 4096, 8
    base = scenario("default", N)
    distinct = [{"Phi0": float(a), "PhiIni": float(b), "PhiNR": float(b), "k3": float(k), "k4": float(k)}
                for a, b, k in zip(np.linspace(0.5, 0.8, D), np.linspace(0.8, 0.5, D), np.logspace(-2, -1, D))]
    order = np.random.default_rng(11).integers(0, D, B)
    eq = LMAHeureuxPorosityDiff.from_scenario(base, device=0, instances=[distinct[i] for i in order])
    eq.use_stream(torch.cuda.current_stream().cuda_stream)
    y0d = np.stack([synthetic_state(base | d, N, amplitude=0.02) for d in distinct])
    dx2 = (eq.Depths.length / N) ** 2
    yd = torch.from_numpy(y0d[order]).cuda()
    res = eq.sweep_rk45_device(yd.data_ptr(), (0.0, 1.0), 0.5 * dx2, 1e-3, 1e-3, max_attempts=40)
    torch.cuda.synchronize()
    got = yd.cpu().numpy()
    for i in range(D):
        rows = np.flatnonzero(order == i)
        assert len(rows) > 100
        assert np.all(got[rows] == got[rows[0]]), i                       # bit-identical replicas
        assert len({(res[r].n_accepted, res[r].n_rejected, res[r].t_reached) for r in rows}) == 1
        yref, st, *_ = oracle.rk45(oracle.params_from_model(eq, int(rows[0])), N, y0d[i], 0.0, 1.0, 0.5 * dx2, 1e-3, 1e-3, max_attempts=40)
        assert (res[rows[0]].status, res[rows[0]].n_accepted, res[rows[0]].n_rejected) == (2, st.n_accepted, st.n_rejected)
        assert rel_to_max(got[rows[0]], yref) <= 1e-9
    eq.close()
