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


def test_assignment_modes_and_the_config4_sweep_over_eight_ranks():
    """BASELINE configs[3]: 32 768 instances over 8 ranks.  Every assignment mode partitions [0, 32 768) exactly; a rank's parameter
    blocks (what marl_ctx_create receives) are the corresponding blocks of the unsharded sweep, byte for byte; a skewed cost model is
    balanced by the cost-sorted assignment and by round robin, not by contiguous ranges."""
    from marlpde_amd.LHeureux_model import instance_kwargs, pack_blocks
    from marlpde_amd.sweep import assign, implicit_cost_proxy, product_grid, shard
    n, world = 32768, 8
    insts = product_grid(Phi0=np.linspace(0.5, 0.8, 32), PhiIni=np.linspace(0.5, 0.8, 32), k3=np.logspace(-2, -1, 32))
    for d in insts:
        d.update(PhiNR=d["PhiIni"], k4=d["k3"])
    assert len(insts) == n
    base = scenario("default", 1024)
    cost = implicit_cost_proxy(base, insts)
    assert cost.min() == 1.0 and cost.max() > 50 * cost.min()
    for mode in ("contiguous", "round_robin", "cost"):
        parts = [assign(n, r, world, mode, cost) for r in range(world)]
        assert np.array_equal(np.sort(np.concatenate(parts)), np.arange(n)), mode
        assert all(p == sorted(p) for p in parts)
        if mode != "cost":
            assert [len(p) for p in parts] == [4096] * 8
    assert [assign(n, r, world) for r in range(world)] == [list(range(*shard(n, r, world))) for r in range(world)]
    load = {mode: np.array([cost[assign(n, r, world, mode, cost)].sum() for r in range(world)]) for mode in ("contiguous", "round_robin", "cost")}
    assert load["contiguous"].max() > 2.0 * load["contiguous"].mean()          # the rank that owns the Phi0 = 0.8 end of the grid
    assert load["round_robin"].max() < 1.02 * load["round_robin"].mean()
    assert load["cost"].max() < 1.001 * load["cost"].mean()
    L = base["max_depth"] / base["Xstar"]
    full = bytes(pack_blocks(instance_kwargs(base, insts), L))
    blk = len(full) // n
    for mode in ("contiguous", "cost"):
        for r in (0, 3, 7):
            mine = assign(n, r, world, mode, cost)
            local = bytes(pack_blocks(instance_kwargs(base, [insts[i] for i in mine]), L))
            assert local == b"".join(full[i * blk:(i + 1) * blk] for i in mine), (mode, r)


def _balanced_sweep_worker(rank, world, N, insts, balance, cost):
    from cpu_engines import OracleSweepEngine
    from marlpde_amd.sweep import assign, run_sweep_radau
    base = scenario("default", N)
    seen = []

    def factory(bp, inst):
        seen.extend(inst)
        return OracleSweepEngine(bp, inst)
    y, status, acc, rej, t = run_sweep_radau(base, insts, (0.0, 0.02), 1e-6, 1e-3, 1e-3, engine_factory=factory, balance=balance, cost=cost)
    return y, status, acc, rej, seen, assign(len(insts), rank, world, balance, cost)


@pytest.mark.parametrize("balance", ["round_robin", "cost"])
def test_balanced_implicit_sweep_returns_results_in_input_order(oracle, balance):
    """run_sweep_radau over 3 ranks with a skewed cost model: every rank integrates exactly the instances assign() gives it, and every
    rank ends with ALL results in the order of `instances` - equal to the oracle instance by instance."""
    from marlpde_amd.sweep import product_grid
    N, world = 24, 3
    insts = product_grid(Phi0=[0.5, 0.55, 0.6, 0.65, 0.7, 0.75, 0.8])
    for d in insts:
        d.update(PhiIni=d["Phi0"], PhiNR=d["Phi0"])
    cost = [1, 1, 1, 1, 2, 9, 30]
    out = _spawn(_balanced_sweep_worker, world, N, insts, balance, cost)
    base = scenario("default", N)
    owners = [out[r][5] for r in range(world)]
    assert sorted(i for o in owners for i in o) == list(range(len(insts)))
    if balance == "cost":
        assert owners[0] == [6] and 5 in owners[1]            # the heaviest alone, the second heaviest on the next rank
    for r in range(world):
        y, status, acc, rej, seen, mine = out[r]
        assert seen == [insts[i] for i in mine]
        assert y.shape == (len(insts), 5 * N) and list(status) == [0] * len(insts)
        for i, inst in enumerate(insts):
            p = base | inst
            y0 = np.repeat([p["CAIni"], p["CCIni"], p["cCaIni"], p["cCO3Ini"], p["PhiIni"]], N)
            yref, st, *_ = oracle.radau(oracle.params_from_dict(p), N, y0, 0.0, 0.02, 1e-6, 1e-3, 1e-3)
            assert np.array_equal(y[i], yref) and (acc[i], rej[i]) == (st.n_accepted, st.n_rejected), (r, i)


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


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_domain_decomposition_matches_single_grid(oracle, world):
    """Ranks exchange 6-cell halos once per attempt and all-gather one record each; every rank takes the same
    accept/reject decisions, and the result equals the single-grid integration.  (8 = the rank count of BASELINE configs[4].)"""
    from marlpde_amd.domain import partition
    assert partition(10, 3) == [(0, 3), (3, 6), (6, 10)]
    N = 50 if world < 8 else 72
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


# ---- transport set-up of the domain-decomposed driver: collectives must pair up whatever fails locally ----------------------
def _dd_transport_worker(rank, world, N, t1, first_step, rtol, atol, fail_where, fail_rank):
    from cpu_engines import OracleSlabEngine
    from marlpde_amd._abi import MarlError
    from marlpde_amd.domain import DomainDecomposedRK45, owned_slice

    class Stub(OracleSlabEngine):
        """Claims the library-side transport under gloo; one of its set-up calls fails on one rank."""
        native_backends = ("gloo",)
        calls = []

        def _maybe_fail(self, where):
            self.calls.append(where)
            if fail_where == where and rank == fail_rank:
                raise MarlError(f"{where} failed on rank {rank} (injected)")

        def comm_id(self):
            self._maybe_fail("comm_id")
            return b"\x01" * 128

        def comm_probe(self):
            self._maybe_fail("comm_probe")

        def comm_init(self, uid, r, w):
            assert uid == b"\x01" * 128 and (r, w) == (rank, world)
            self._maybe_fail("comm_init")

        def run(self):
            raise AssertionError("the library loop must not be entered after a failed set-up")

    p = scenario("A", N)
    y0 = synthetic_state(p, N, amplitude=0.05)
    dd = DomainDecomposedRK45(p, N, engine_factory=lambda b, e: Stub(p, N, b, e, 6), poll=5)
    y = torch.from_numpy(owned_slice(y0, N, dd.begin, dd.end))
    st = dd.integrate(y, (0.0, t1), first_step, rtol, atol, 0)
    return y.numpy(), (st.status, st.n_accepted, st.n_rejected), dd.native, dd.transport, list(Stub.calls)


@pytest.mark.parametrize("fail_where,fail_rank", [("comm_id", 0), ("comm_probe", 1), ("comm_init", 1)])
def test_domain_transport_setup_failure_on_one_rank_falls_back_everywhere(oracle, fail_where, fail_rank):
    """when rank 0 cannot make the RCCL id (or one rank cannot load RCCL, or its comm_init fails) every rank must
    still walk through the same collectives and end on the same transport - here: all fall back to the host loop over
    torch.distributed, and the run equals the single-grid integration.  Before the fix rank 0 skipped the broadcast the other
    ranks were waiting in."""
    N = 40
    p = scenario("A", N)
    dx2 = ((p["max_depth"] / p["Xstar"]) / N) ** 2
    t1, h0, rtol, atol = 12 * dx2, 0.4 * dx2, 1e-4, 1e-6
    y0 = synthetic_state(p, N, amplitude=0.05)
    yref, st, *_ = oracle.rk45(oracle.params_from_dict(p), N, y0, 0.0, t1, h0, rtol, atol)
    out = _spawn(_dd_transport_worker, 2, N, t1, h0, rtol, atol, fail_where, fail_rank)
    assert [out[r][2] for r in (0, 1)] == [False, False]
    assert all("host loop" in out[r][3] for r in (0, 1))
    assert {out[r][1] for r in (0, 1)} == {(0, st.n_accepted, st.n_rejected)}
    got = np.concatenate([out[r][0].reshape(5, -1) for r in (0, 1)], axis=1)
    assert np.max(np.abs(got - yref.reshape(5, N))) <= 1e-13
    if fail_where == "comm_id":       # nobody may enter the (collective) comm_init without an id
        assert all("comm_init" not in out[r][4] for r in (0, 1))
    if fail_where == "comm_probe":    # agreed on before anyone enters comm_init
        assert all("comm_init" not in out[r][4] for r in (0, 1))


def test_slab_comm_entry_points_fail_fast_without_a_context_or_library():
    """What can be said about the RCCL set-up without a GPU: the probe is local and returns (it never blocks), a missing
    context is an error, not a crash."""
    import ctypes as C
    from marlpde_amd import _abi
    lib = _abi.load()
    assert lib.marl_slab_comm_init(None, None, b"\x00" * 128, 0, 2) == -1
    rc = lib.marl_slab_comm_probe(b"/nonexistent/librccl.so")
    assert rc in (0, -2)      # 0: the loader found a default librccl; -2: none - either way an answer, at once
    if rc != 0:
        assert b"librccl" in lib.marl_last_error(None)
    buf = C.create_string_buffer(128)
    assert lib.marl_slab_comm_id(None, None) == -1 and buf.raw == b"\x00" * 128
