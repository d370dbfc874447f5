"""One very large depth grid across several GPUs: 1-D domain decomposition with halo exchange.

Not in the reference (it never decomposes the depth axis) - this is BASELINE config 5.
One process per GPU (``torch.distributed``; backend "nccl" is RCCL over xGMI on ROCm).  Rank r owns the
contiguous cells [N r / P, N (r+1) / P); its slab buffer carries ``halo`` cells of each neighbour.  A fused
Dormand-Prince attempt consumes 6 cells of halo per side (marl_kernels.h), so ONE exchange per attempt
suffices: 2 arrays (y_new, f_new) x 5 fields x halo cells = 480 B per side.  The step controller needs the
global error norm: each rank reduces its cells to one 8-double record.  Record and both strips travel in ONE
all-gather of ~1 KB per rank and attempt (pure latency; one collective instead of a batch of neighbour
sends/receives plus an all-gather); every rank reads its neighbours' strips out of the gathered buffer and
combines the records IN RANK ORDER, so all ranks take bit-identical accept/reject decisions.  No host synchronisation inside the loop: kernels and collectives are enqueued on one stream, the
controller state lives on the device, and the host only polls the status every ``poll`` attempts.

The arithmetic is behind a small engine interface so that the driver logic is testable on CPU (gloo) with a
reference engine injected by the tests; the product engine is :class:`HipSlabEngine` (C ABI, marl_slab_*).
"""
import ctypes as C
import os

import numpy as np

from . import _abi

HALO = 6          # cells per side consumed by one fused attempt
STRIP = 2 * 5     # (y, f) x five fields


def partition(N, world):
    """Contiguous, near-even split of N cells: list of (begin, end) per rank."""
    edges = [(N * r) // world for r in range(world + 1)]
    return [(edges[r], edges[r + 1]) for r in range(world)]


class HipSlabEngine:
    """A rank's slab on its GPU (marl_ctx_create_slab and the marl_slab_* calls)."""

    native_backends = ("nccl",)   # torch.distributed backends under which the library-side RCCL transport is attempted

    def __init__(self, pde_parms, N_global, begin, end, device, halo=HALO):
        import torch
        from .LHeureux_model import LMAHeureuxPorosityDiff  # noqa: F401  (packing helper below mirrors its ctor)
        self.torch = torch
        self.lib = _abi.load()
        self.ctx = C.c_void_p()
        blk = _abi.MarlParams()
        for name in _abi.PARAM_DOUBLES[:30]:
            setattr(blk, name, float(pde_parms[name]))
        blk.length = float(pde_parms["max_depth"] / pde_parms["Xstar"])
        blk.shallow_limit = float(pde_parms["ShallowLimit"]) / float(pde_parms["Xstar"])
        blk.deep_limit = float(pde_parms["DeepLimit"]) / float(pde_parms["Xstar"])
        blk.FV_switch = int(pde_parms["FV_switch"])
        blk.dPhi_variable = int(bool(pde_parms.get("dPhi_variable", False)))
        rc = self.lib.marl_ctx_create_slab(C.byref(blk), N_global, begin, end, halo, device, C.byref(self.ctx))
        if rc != 0:
            self.ctx = C.c_void_p()
            _abi.check(None, rc, "marl_ctx_create_slab")
        self.device = torch.device("cuda", device)
        self.n_own = end - begin
        _abi.check(self.ctx, self.lib.marl_set_stream(self.ctx, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)),
                   "marl_set_stream")
        _abi.check(self.ctx, self.lib.marl_set_option(self.ctx, b"poll_interval", 32), "marl_set_option")
        for name, value in _abi.lab_options():   # (kernel-lab A/B runs only: MARL_HIP_OPTIONS)
            _abi.check(self.ctx, self.lib.marl_set_option(self.ctx, name.encode(), int(value)), "marl_set_option")

    def _call(self, name, *args):
        _abi.check(self.ctx, getattr(self.lib, name)(self.ctx, *args), name)

    @staticmethod
    def _p(t):
        return None if t is None else C.c_void_p(t.data_ptr())

    def new_tensor(self, n):
        return self.torch.zeros(n, dtype=self.torch.float64, device=self.device)

    def load(self, y_owned):
        self._call("marl_slab_load", self._p(y_owned))

    def store(self, y_owned):
        self._call("marl_slab_store", self._p(y_owned))

    def pack(self, which, send_lo, send_hi):
        self._call("marl_slab_pack", which, self._p(send_lo), self._p(send_hi))

    def unpack(self, which, recv_lo, recv_hi):
        self._call("marl_slab_unpack", which, self._p(recv_lo), self._p(recv_hi))

    def rhs0(self):
        self._call("marl_slab_rhs0")

    def monitors(self, rec):
        self._call("marl_slab_monitors", self._p(rec))

    def init_control(self, recs, nrec, t0, t1, first_step, rtol, atol, max_attempts):
        self._call("marl_slab_init_control", self._p(recs), nrec, t0, t1, first_step, rtol, atol, max_attempts)

    def attempt(self, rec):
        self._call("marl_slab_attempt", self._p(rec))

    def control(self, recs, nrec):
        self._call("marl_slab_control", self._p(recs), nrec)

    def status(self):
        st = _abi.MarlStats()
        self._call("marl_slab_status", C.byref(st))
        return st

    # -- the exchange inside the library (RCCL through the C ABI; no Python in the per-attempt path) ------------------
    @staticmethod
    def rccl_path():
        """The librccl this process already has (PyTorch-ROCm bundles one): the library dlopens THAT one."""
        import torch
        p = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        return p.encode() if os.path.exists(p) else None

    def comm_id(self):
        buf = C.create_string_buffer(128)
        _abi.check(None, self.lib.marl_slab_comm_id(self.rccl_path(), buf), "marl_slab_comm_id")
        return buf.raw

    def comm_probe(self):
        """Local check that the RCCL entry points can be loaded - everything about comm_init that is NOT collective."""
        _abi.check(None, self.lib.marl_slab_comm_probe(self.rccl_path()), "marl_slab_comm_probe")

    def comm_init(self, uid, rank, world):
        self._call("marl_slab_comm_init", self.rccl_path(), uid, rank, world)

    def exchange(self, which):
        self._call("marl_slab_exchange", which)

    def monitors_into_message(self):
        self._call("marl_slab_monitors", None)

    def init_control_gathered(self, nrec, t0, t1, first_step, rtol, atol, max_attempts):
        self._call("marl_slab_init_control", None, nrec, t0, t1, first_step, rtol, atol, max_attempts)

    def run(self):
        st = _abi.MarlStats()
        self._call("marl_slab_run", C.byref(st))
        return st

    def close(self):
        if self.ctx:
            self.lib.marl_ctx_destroy(self.ctx)
            self.ctx = C.c_void_p()


class DomainDecomposedRK45:
    """Adaptive RK45 of one N-cell grid over all ranks of ``group``.

    ``engine_factory(begin, end) -> engine`` builds the rank's slab engine (default: HipSlabEngine on
    ``cuda:LOCAL_RANK``)."""

    def __init__(self, pde_parms, N, group=None, device=None, engine_factory=None, halo=HALO, poll=16):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.N = int(N)
        self.halo = halo
        self.poll = poll
        self.parts = partition(self.N, self.world)
        self.begin, self.end = self.parts[self.rank]
        if min(e - b for b, e in self.parts) < halo:
            raise ValueError(f"N = {N} is too small for {self.world} slabs with a halo of {halo} cells")
        if engine_factory is None:
            # one process per GPU: the device is the LOCAL rank (a global rank only equals it on the first node)
            dev = int(os.environ.get("LOCAL_RANK", self.rank)) if device is None else device
            engine_factory = lambda b, e: HipSlabEngine(pde_parms, self.N, b, e, dev, halo)  # noqa: E731
        self.engine = engine_factory(self.begin, self.end)
        e = self.engine
        n = STRIP * halo
        # One message per rank and attempt: [record (8) | lower strip (n) | upper strip (n)].  A single all-gather moves
        # the halos AND the step-control records (P x ~1 KB: pure latency either way, and one collective instead of a
        # batch of sends/receives plus an all-gather); every rank then reads its two neighbours' strips in place.
        self.msg = 8 + 2 * n
        self.send = e.new_tensor(self.msg)
        self.rec, self.send_lo, self.send_hi = self.send[:8], self.send[8:8 + n], self.send[8 + n:]
        self.gathered = e.new_tensor(self.msg * self.world).view(self.world, self.msg)
        lo_src = self.gathered[self.rank - 1] if self.rank > 0 else self.send                 # lower neighbour's UPPER strip
        hi_src = self.gathered[self.rank + 1] if self.rank < self.world - 1 else self.send    # upper neighbour's LOWER strip
        self.recv_lo, self.recv_hi = lo_src[8 + n:], hi_src[8:8 + n]
        self.recs = e.new_tensor(8 * self.world)
        # Transport.  Product engine + (one slab | RCCL): the exchange and the whole per-attempt loop run inside the library
        # (marl_slab_run: ncclAllGather on the context's stream, no Python per attempt).  Otherwise (gloo / injected test
        # engines): the same sequence driven from here with torch.distributed collectives.
        self.native = False
        uid = None
        backend = dist.get_backend(group) if dist.is_initialized() else None
        want = os.environ.get("MARL_DD_TRANSPORT", "auto")          # auto | native | torch | rccl1 (one-rank communicator)
        # Every rank walks through the SAME sequence of collectives whatever fails locally (a rank that skipped the broadcast and
        # went on to the all-reduce would pair different collectives or hang the job): eligibility depends only on values that
        # are equal on all ranks; failures are carried as data (uid None / ok 0) into the next collective, never as control flow
        # around it.  ncclCommInitRank itself is collective - everything that can be checked locally (library loadable, symbols
        # present: comm_probe) is agreed on BEFORE it is entered.
        eligible = (getattr(e, "native_backends", None) is not None and want != "torch"
                    and (self.world == 1 or backend in e.native_backends))
        error = None
        if eligible:
            need_comm = self.world > 1 or want == "rccl1"
            ok = 1
            if need_comm:
                if self.rank == 0:
                    try:
                        uid = e.comm_id()
                    except _abi.MarlError as ex:
                        error = ex
                try:
                    e.comm_probe()
                except _abi.MarlError as ex:
                    error, ok = ex, 0
                if self.world > 1:
                    box = [uid]
                    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
                    uid = box[0]
                    ok = self._all_min(ok if uid is not None else 0, e, backend)
                elif uid is None:
                    ok = 0
            if ok:
                try:
                    e.comm_init(uid, self.rank, self.world)
                except _abi.MarlError as ex:
                    error, ok = ex, 0
            if self.world > 1:
                ok = self._all_min(ok, e, backend)
            self.native = bool(ok)
            if not self.native and want in ("native", "rccl1"):
                raise error if error is not None else _abi.MarlError(f"MARL_DD_TRANSPORT={want}: the library-side transport failed on another rank")
        self.transport = ("library loop, one slab (no communicator)" if self.native and uid is None else
                          "library loop, ncclAllGather (RCCL via the C ABI)" if self.native else
                          "none (one slab), host loop" if self.world == 1 else
                          f"host loop, torch.distributed.all_gather_into_tensor, backend {backend}")

    # -- communication ---------------------------------------------------------------------------
    def _all_min(self, value, e, backend):
        """MIN of one int over the group (every rank calls it at the same point of the transport set-up)."""
        flag = e.new_tensor(1).fill_(float(value)) if backend == "nccl" else __import__("torch").tensor([float(value)], dtype=__import__("torch").float64)
        self.dist.all_reduce(flag, op=self.dist.ReduceOp.MIN, group=self.group)
        return int(flag.item())

    def _exchange(self, which):
        """pack -> all-gather -> unpack; afterwards ``recs`` holds every rank's record in rank order."""
        self.engine.pack(which, self.send_lo, self.send_hi)
        if self.world == 1:
            self.recs.copy_(self.rec)
            return
        self.dist.all_gather_into_tensor(self.gathered.view(-1), self.send, group=self.group)
        self.recs.view(self.world, 8).copy_(self.gathered[:, :8])
        self.engine.unpack(which, self.recv_lo, self.recv_hi)

    # -- the integration ---------------------------------------------------------------------------
    def integrate(self, y_owned, t_span, first_step, rtol, atol, max_attempts=0):
        """``y_owned``: this rank's cells, tensor [5 * n_own] (field-major), advanced in place.
        Returns the marl_stats of the run (identical on every rank)."""
        e = self.engine
        if self.native:
            e.load(y_owned)
            e.exchange(0)                # y halos
            e.rhs0()
            e.monitors_into_message()
            e.exchange(0)                # f halos (FSAL vector) + the monitors record of y(t0)
            e.init_control_gathered(self.world, float(t_span[0]), float(t_span[1]), float(first_step), float(rtol), float(atol),
                                    int(max_attempts))
            st = e.run()                 # attempt -> reduce + pack -> all-gather -> unpack + control, inside the library
            e.store(y_owned)
            return st
        e.load(y_owned)
        self._exchange(0)            # y halos
        e.rhs0()
        e.monitors(self.rec)
        self._exchange(0)            # f halos (FSAL vector) + the monitors record of y(t0)
        e.init_control(self.recs, self.world, float(t_span[0]), float(t_span[1]), float(first_step), float(rtol),
                       float(atol), int(max_attempts))
        executed = 0
        while True:
            # never more attempts per batch than the budget has left (further ones would be dispatches that do nothing)
            batch = self.poll if max_attempts <= 0 else max(1, min(self.poll, int(max_attempts) - executed))
            for _ in range(batch):
                e.attempt(self.rec)
                self._exchange(-1)
                e.control(self.recs, self.world)
            st = e.status()
            if st.status != 1:
                break
            executed = int(st.n_accepted + st.n_rejected)
        e.store(y_owned)
        return st

    def close(self):
        self.engine.close()


def owned_slice(y_global, N, begin, end):
    """The [5 * (end-begin)] field-major piece of a global field-major state (host helper)."""
    return np.ascontiguousarray(np.asarray(y_global).reshape(5, N)[:, begin:end]).ravel()
