"""The N > 1 paths under gloo, world_size 2, on CPU: sweep sharding (no data-path collective) and the
domain-decomposed RK45 driver (neighbour halo exchange + all-gathered step control).  The arithmetic is
supplied by oracle-based test doubles (tests/cpu_engines.py); what is under test is the product's driver
logic in marlpde_amd/sweep.py and marlpde_amd/domain.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from common import scenario, synthetic_state


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _spawn(fn, world, *args):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_entry, args=(fn, r, world, port, q, args)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        out = [q.get(timeout=240) for _ in range(world)]      # a crashed rank must fail the test, not hang it
    finally:
        for p in procs:
            p.join(30)
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs)
    return dict(out)


def _entry(fn, rank, world, port, q, args):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here), here]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank, fn(rank, world, *args)))
    finally:
        dist.destroy_process_group()


# ---- sweep -----------------------------------------------------------------------------------------
def _sweep_worker(rank, world, N, insts):
    from cpu_engines import OracleSweepEngine
    from marlpde_amd.sweep import run_sweep_rk45, shard
    base = scenario("default", N)
    dx2 = ((base["max_depth"] / base["Xstar"]) / N) ** 2
    y, status, acc, rej, t = run_sweep_rk45(base, insts, (0.0, 1.0), 0.5 * dx2, 1e-3, 1e-3, max_attempts=25,
                                            engine_factory=lambda bp, inst: OracleSweepEngine(bp, inst))
    return y, status, acc, rej, t, shard(len(insts), rank, world)


def test_sweep_shards_without_collectives_and_gathers(oracle):
    from marlpde_amd.sweep import product_grid, shard
    assert [shard(7, r, 3) for r in range(3)] == [(0, 2), (2, 4), (4, 7)]
    N = 32
    insts = product_grid(Phi0=[0.55, 0.7, 0.8], k3=[0.02, 0.1])
    for d in insts:
        d.update(PhiIni=d["Phi0"], PhiNR=d["Phi0"], k4=d["k3"])
    assert len(insts) == 6 and insts[1]["k3"] == 0.1 and insts[2]["Phi0"] == 0.7
    out = _spawn(_sweep_worker, 2, N, insts)
    assert out[0][5] == (0, 3) and out[1][5] == (3, 6)
    base = scenario("default", N)
    dx2 = ((base["max_depth"] / base["Xstar"]) / N) ** 2
    for r in (0, 1):
        y, status, acc, rej, t, _ = out[r]
        assert y.shape == (6, 5 * N) and list(status) == [2] * 6
        for i, inst in enumerate(insts):
            p = base | inst
            y0 = np.repeat([p["CAIni"], p["CCIni"], p["cCaIni"], p["cCO3Ini"], p["PhiIni"]], N)
            yref, st, *_ = oracle.rk45(oracle.params_from_dict(p), N, y0, 0.0, 1.0, 0.5 * dx2, 1e-3, 1e-3, max_attempts=25)
            assert np.array_equal(y[i], yref) and (acc[i], rej[i]) == (st.n_accepted, st.n_rejected) and t[i] == st.t


def _radau_sweep_worker(rank, world, N, insts):
    from cpu_engines import OracleSweepEngine
    from marlpde_amd.sweep import run_sweep_radau, shard
    base = scenario("default", N)
    y, status, acc, rej, t = run_sweep_radau(base, insts, (0.0, 0.05), 1e-6, 1e-3, 1e-3,
                                             engine_factory=lambda bp, inst: OracleSweepEngine(bp, inst))
    return y, status, acc, rej, t, shard(len(insts), rank, world)


def test_radau_sweep_shards_without_collectives_and_gathers(oracle):
    """The same sharding for the reference's default solver (run_sweep_radau): 5 scenarios over 2 ranks, every rank ends with all results."""
    from marlpde_amd.sweep import product_grid
    N = 32
    insts = product_grid(Phi0=[0.5, 0.6, 0.7, 0.75, 0.8])
    for d in insts:
        d.update(PhiIni=d["Phi0"], PhiNR=d["Phi0"])
    out = _spawn(_radau_sweep_worker, 2, N, insts)
    assert out[0][5] == (0, 2) and out[1][5] == (2, 5)
    base = scenario("default", N)
    for r in (0, 1):
        y, status, acc, rej, t, _ = out[r]
        assert y.shape == (5, 5 * N) and list(status) == [0] * 5 and np.all(t == 0.05)
        for i, inst in enumerate(insts):
            p = base | inst
            y0 = np.repeat([p["CAIni"], p["CCIni"], p["cCaIni"], p["cCO3Ini"], p["PhiIni"]], N)
            yref, st, *_ = oracle.radau(oracle.params_from_dict(p), N, y0, 0.0, 0.05, 1e-6, 1e-3, 1e-3)
            assert np.array_equal(y[i], yref) and (acc[i], rej[i]) == (st.n_accepted, st.n_rejected)


# ---- domain decomposition ----------------------------------------------------------------------------
def _dd_worker(rank, world, N, t1, first_step, rtol, atol, max_attempts):
    from cpu_engines import OracleSlabEngine
    from marlpde_amd.domain import DomainDecomposedRK45, owned_slice
    p = scenario("A", N)
    y0 = synthetic_state(p, N, amplitude=0.05)
    dd = DomainDecomposedRK45(p, N, engine_factory=lambda b, e: OracleSlabEngine(p, N, b, e, 6), poll=5)
    y = torch.from_numpy(owned_slice(y0, N, dd.begin, dd.end))
    st = dd.integrate(y, (0.0, t1), first_step, rtol, atol, max_attempts)
    return y.numpy(), (st.status, st.n_accepted, st.n_rejected, st.nfev, st.t), (dd.begin, dd.end)


@pytest.mark.parametrize("world", [1, 2, 3])
def test_domain_decomposition_matches_single_grid(oracle, world):
    """Ranks exchange 6-cell halos once per attempt and all-gather one record each; every rank takes the same
    accept/reject decisions, and the result equals the single-grid integration."""
    from marlpde_amd.domain import partition
    assert partition(10, 3) == [(0, 3), (3, 6), (6, 10)]
    N = 50
    p = scenario("A", N)
    dx2 = ((p["max_depth"] / p["Xstar"]) / N) ** 2
    t1, h0, rtol, atol = 30 * dx2, 0.4 * dx2, 1e-4, 1e-6
    y0 = synthetic_state(p, N, amplitude=0.05)
    yref, st, *_ = oracle.rk45(oracle.params_from_dict(p), N, y0, 0.0, t1, h0, rtol, atol)
    assert st.n_rejected > 0, "the test run must contain rejected attempts"
    out = _spawn(_dd_worker, world, N, t1, h0, rtol, atol, 0)
    stats = {out[r][1] for r in range(world)}
    assert stats == {(0, st.n_accepted, st.n_rejected, st.nfev, t1)}
    got = np.concatenate([out[r][0].reshape(5, -1) for r in range(world)], axis=1)
    assert np.max(np.abs(got - yref.reshape(5, N))) <= 1e-13
    # an attempt budget stops every rank at the same attempt
    out = _spawn(_dd_worker, world, N, t1, h0, rtol, atol, 7)
    _, st2, *_ = oracle.rk45(oracle.params_from_dict(p), N, y0, 0.0, t1, h0, rtol, atol, max_attempts=7)
    assert {out[r][1][:3] for r in range(world)} == {(2, st2.n_accepted, st2.n_rejected)}
