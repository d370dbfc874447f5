"""Batched parameter sweeps: many independent model instances, sharded over the GPUs of a node.

Not in the reference (one scenario per process); BASELINE configs 3-4.  Instances are independent - own
parameters, own step-size controller - so ranks take contiguous index ranges and never communicate in the
data path ("replicas of the kernel").  On each GPU one workgroup integrates one instance
entirely on-chip (marl_kernels.h, rk45_sweep_kernel).  Results can be gathered to every rank afterwards
(control plane only).
"""
import os

import numpy as np


def shard(n_instances, rank, world):
    """Contiguous near-even split: the (begin, end) instance range of ``rank``."""
    return (n_instances * rank) // world, (n_instances * (rank + 1)) // world


def product_grid(**axes):
    """Cartesian product of named parameter axes -> list of override dicts (last axis fastest)."""
    names = list(axes)
    grids = np.meshgrid(*[np.asarray(axes[k], dtype=float) for k in names], indexing="ij")
    return [dict(zip(names, (float(g.flat[i]) for g in grids))) for i in range(grids[0].size)]


class HipSweepEngine:
    """The product engine: LMAHeureuxPorosityDiff with a batch of instances on one GPU."""

    def __init__(self, base_parms, instances, device):
        import torch
        from .LHeureux_model import LMAHeureuxPorosityDiff
        self.torch = torch
        self.device = torch.device("cuda", device)
        self.base_parms, self.instances = dict(base_parms), [dict(i) for i in instances]
        self.model = LMAHeureuxPorosityDiff.from_scenario(base_parms, device=device, instances=instances)
        self.model.use_stream(torch.cuda.current_stream(self.device).cuda_stream)

    def integrate_rk45(self, y0, t_span, first_step, rtol, atol, max_attempts):
        """y0: (n_local, 5N) host array.  Returns (y_final (n_local, 5N), list of RK45Result)."""
        yd = self.torch.from_numpy(np.ascontiguousarray(y0)).to(self.device)
        res = self.model.sweep_rk45_device(yd.data_ptr(), t_span, first_step, rtol, atol, max_attempts)
        return yd.cpu().numpy(), res

    def integrate_radau(self, y0, t_span, first_step, rtol, atol, max_attempts):
        """The reference's default solver over the shard: every instance its own Radau step logic, advanced together on the device
        (marl_sweep_radau_dev).  Returns (y_final (n_local, 5N), list of RK45Result)."""
        yd = self.torch.from_numpy(np.ascontiguousarray(y0)).to(self.device)
        res = self.model.sweep_radau_device(yd.data_ptr(), t_span, first_step, rtol, atol, max_attempts)
        return yd.cpu().numpy(), res

    def integrate_bdf(self, y0, t_span, first_step, rtol, atol, max_attempts, workers=None):
        """scipy BDF semantics (the other implicit method of the reference's Solver, marlpde/parameters.py:205-219) for every instance of
        the shard.  There is no batched BDF kernel set (VERDICT r2 missing #5 called it minor): the instances run as concurrent SINGLE runs
        (marl_integrate_bdf, one context and one HIP stream per worker thread - ctypes releases the GIL, and a single BDF run of a small
        grid leaves the GPU mostly idle between its small launches), so each instance is bit for bit the single run.  Returns
        (y_final (n_local, 5N), list of RK45Result)."""
        from concurrent.futures import ThreadPoolExecutor
        from .LHeureux_model import LMAHeureuxPorosityDiff
        torch = self.torch
        y0 = np.ascontiguousarray(y0, dtype=np.float64)
        n_inst = len(self.instances)
        workers = max(1, min(n_inst, workers or min(16, os.cpu_count() or 1)))
        out = [None] * n_inst
        yf = np.empty_like(y0)

        def work(w):
            stream = torch.cuda.Stream(device=self.device)
            one = None
            try:
                for b in range(w, n_inst, workers):
                    p = self.base_parms | self.instances[b]
                    if one is None:   # one context per worker; later scenarios only replace its constants (marl_ctx_set_params)
                        one = LMAHeureuxPorosityDiff.from_scenario(p, device=self.device.index)
                        one.use_stream(stream.cuda_stream)
                    else:
                        one.set_scenario(p)
                    r = one.integrate_bdf(y0[b], t_span, first_step, rtol, atol, max_attempts=max_attempts)
                    out[b] = r
                    yf[b] = r.y_final
            finally:
                if one is not None:
                    one.close()

        with ThreadPoolExecutor(max_workers=workers) as pool:
            for f in [pool.submit(work, w) for w in range(workers)]:
                f.result()
        return yf, out

    def integrate_rk4(self, y0, dt, nsteps):
        yd = self.torch.from_numpy(np.ascontiguousarray(y0)).to(self.device)
        self.model.sweep_rk4_device(yd.data_ptr(), dt, nsteps)
        self.torch.cuda.synchronize(self.device)
        return yd.cpu().numpy()

    def close(self):
        self.model.close()


def initial_states(base_parms, instances):
    """Uniform initial state per instance (marlpde/Evolve_scenario.py:76-86), shape (n, 5N)."""
    N = int(base_parms["N"])
    out = np.empty((len(instances), 5 * N))
    for i, inst in enumerate(instances):
        p = base_parms | inst
        out[i] = np.repeat([p["CAIni"], p["CCIni"], p["cCaIni"], p["cCO3Ini"], p["PhiIni"]], N)
    return out


def run_sweep_radau(base_parms, instances, t_span, first_step, rtol, atol, max_attempts=0, y0=None, group=None,
                    device=None, engine_factory=None, gather=True):
    """As :func:`run_sweep_rk45` with the reference's DEFAULT solver (scipy Radau semantics, marlpde/parameters.py:213): what the
    reference does one scenario per process (its tests loop over scenarios), sharded over the ranks with no data-path collective."""
    return _run_sweep("integrate_radau", base_parms, instances, t_span, first_step, rtol, atol, max_attempts, y0, group, device, engine_factory, gather)


def run_sweep_bdf(base_parms, instances, t_span, first_step, rtol, atol, max_attempts=0, y0=None, group=None,
                  device=None, engine_factory=None, gather=True):
    """As :func:`run_sweep_radau` with scipy's BDF semantics; on a GPU the shard's instances run as concurrent single runs
    (:meth:`HipSweepEngine.integrate_bdf`)."""
    return _run_sweep("integrate_bdf", base_parms, instances, t_span, first_step, rtol, atol, max_attempts, y0, group, device, engine_factory, gather)


def run_sweep_rk45(base_parms, instances, t_span, first_step, rtol, atol, max_attempts=0, y0=None, group=None,
                   device=None, engine_factory=None, gather=True):
    """Integrate every instance with adaptive RK45; ranks of ``group`` each take a contiguous shard.

    Returns ``(y_final, status, n_accepted, n_rejected, t_reached)`` - for ALL instances when ``gather``,
    otherwise for the local shard.  ``engine_factory(base_parms, local_instances) -> engine`` is the test hook;
    the default is :class:`HipSweepEngine` on ``cuda:rank``."""
    return _run_sweep("integrate_rk45", base_parms, instances, t_span, first_step, rtol, atol, max_attempts, y0, group, device, engine_factory, gather)


def _run_sweep(method, base_parms, instances, t_span, first_step, rtol, atol, max_attempts, y0, group, device, engine_factory, gather):
    import torch.distributed as dist
    on = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank(group) if on else 0
    world = dist.get_world_size(group) if on else 1
    lo, hi = shard(len(instances), rank, world)
    local = list(instances[lo:hi])
    if y0 is None:
        y0 = initial_states(base_parms, instances)
    if engine_factory is None:
        dev = int(os.environ.get("LOCAL_RANK", rank)) if device is None else device   # one process per GPU
        engine_factory = lambda bp, inst: HipSweepEngine(bp, inst, dev)  # noqa: E731
    engine = engine_factory(base_parms, local)
    y, res = getattr(engine, method)(np.asarray(y0)[lo:hi], t_span, first_step, rtol, atol, max_attempts)
    engine.close()
    summary = np.array([[r.status, r.n_accepted, r.n_rejected, r.t_reached] for r in res], dtype=float).reshape(len(local), 4)
    if gather and world > 1:
        parts = [None] * world
        dist.all_gather_object(parts, (y, summary), group=group)
        y = np.concatenate([p[0] for p in parts])
        summary = np.concatenate([p[1] for p in parts])
    return y, summary[:, 0].astype(int), summary[:, 1].astype(int), summary[:, 2].astype(int), summary[:, 3]
