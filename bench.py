#!/usr/bin/env python3
"""Throughput of the hot path on MI355X: grid-point-steps/s of the fused five-field RK integrators.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload rk4_single|sweep_rk45|rk45_single|dd_rk45] ...

Default (N=1): BASELINE.json's headline - ONE grid of 2^20 depth cells, fixed-step classical RK4, fp64,
fused HIP stencil (configs[1] at the size the north_star quotes its target on).  A "step" is one RK4
step of the whole grid.

--gpus N: one process per GPU.  Under `torch.distributed.run` (WORLD_SIZE set) this process IS one rank;
started directly (`python bench.py --gpus N`) it starts the N rank processes itself BEFORE anything touches
the GPU and waits for them.  Rank r binds cuda:LOCAL_RANK and joins an RCCL ("nccl") process group.  The
headline line is the same workload on every rank (each rank integrates its own grid: weak scaling, no
data-path collective); BASELINE configs[2]/[3] (the batched sweep, 4096 instances per GPU, no collective) and
configs[4] (ONE grid of 2^22 cells domain-decomposed over all ranks, RCCL all-gather per attempt) are timed
afterwards and reported under `extra`, best effort and under a deadline (a failure or hang there cannot take
the headline line with it).

Timing: W untimed warm-up steps, then blocks of EXACTLY K steps, each bracketed by barrier +
torch.cuda.synchronize() on both sides; the block is repeated until >= 50 ms have been timed and the MEDIAN
block (max over ranks per block) is reported (`reps`), so that a small K measures the kernel and not one
launch/synchronise round trip.

One JSON line is printed by rank 0 (contract in the task description); `roofline` is for the dominant
kernel with HIP-event timing on the launch stream, `cpu_baseline` is the oracle (a C port of the
reference's serial loop) timed on this host on a bounded sample of the same workload.
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
# fp64 vector issue ceiling: 256 CUs x 4 SIMDs, one wave64 fp64 VALU instruction per 4 cycles and SIMD, at the 2400 MHz maximum
# clock of the same guide (= 78.6 TFLOP/s of fp64 FMA).  The chip holds 1.9 - 2.35 GHz under this load: the fraction is
# against the spec clock, as the HBM fraction is against the spec bandwidth.
VALU_PEAK_WAVE_INSTS_PER_S = 256 * 4 * 2.4e9 / 4.0
BYTES_PER_POINT_STEP = 80.0    # 5 fields x 8 B read + 5 x 8 B written per grid-point-step
MIN_TIMED_S = 0.05             # repeat the K-step block until this much has been timed
MAX_REPS = 400
EXTRAS_DEADLINE_S = float(os.environ.get("MARL_BENCH_EXTRAS_DEADLINE", "240"))


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="rk4_single", choices=["rk4_single", "sweep_rk45", "sweep_rk4", "rk45_single", "dd_rk45"])
    ap.add_argument("--n", type=int, default=None, help="cells per grid (default 2^20 single, 1024 sweep, 2^22 dd)")
    ap.add_argument("--batch", type=int, default=4096, help="instances per GPU for the sweep workloads")
    ap.add_argument("--layout", type=int, default=1, help="device layout of single-grid runs: 0 field-major, 1 tiled")
    ap.add_argument("--variant", type=int, default=-1, help="kernel variant (see DESIGN.md); -1 = default")
    ap.add_argument("--no-reuse", action="store_true", help="rk4_single: every evaluation takes the full transcendental path")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline workload only")
    return ap.parse_args(argv)


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N rank processes (this parent never touches the GPU)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        while any(p.poll() is None for p in procs):
            for p in procs:
                if p.poll() not in (None, 0):   # one rank failed: the others would wait in a collective for ever
                    rc = p.returncode
                    for q in procs:
                        if q.poll() is None:
                            q.terminate()
            time.sleep(0.2)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc or max((p.returncode or 0) for p in procs)


def synthetic(p, N):
    """SURVEY.md 8d synthetic input: initial values x (1 + 0.01 sin(2 pi 8 x / L)); no RNG."""
    import numpy as np
    L = p["max_depth"] / p["Xstar"]
    x = (np.arange(N) + 0.5) * (L / N)
    y = np.stack([np.full(N, p[k]) for k in ("CAIni", "CCIni", "cCaIni", "cCO3Ini", "PhiIni")])
    return (y * (1.0 + 0.01 * np.sin(2 * np.pi * 8 * x / L))).ravel()


def main():
    args = parse_args()
    if args.gpus < 1:
        sys.exit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; "
                 "run `python bench.py --gpus N` (it starts the ranks itself) or torchrun --nproc-per-node N ... --gpus N")

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool (RCCL / tensor sharing)
    # stdout carries exactly ONE line, the JSON record: everything else that native libraries print there (RCCL's version
    # banner at communicator creation, for one) is sent to stderr by pointing fd 1 at fd 2 until the record is written
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import numpy as np
    import torch
    import torch.distributed as dist
    from dataclasses import asdict
    from marlpde_amd import _abi
    from marlpde_amd.LHeureux_model import LMAHeureuxPorosityDiff
    from marlpde_amd.parameters import Map_Scenario

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the package has no CPU path)"
    # rehearsal of the N > 1 path on a one-GPU box: MARL_BENCH_BACKEND=gloo MARL_BENCH_ONE_DEVICE=1 (all ranks on cuda:0)
    if os.environ.get("MARL_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or os.environ.get("MARL_BENCH_FORCE_DIST") == "1"   # FORCE_DIST: one-rank RCCL group (init + collectives run)
    backend = os.environ.get("MARL_BENCH_BACKEND", "nccl")   # "nccl" is RCCL on ROCm
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    stream = torch.cuda.current_stream()
    red_dev = "cuda" if (not use_dist or backend == "nccl") else "cpu"

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(values):
        t = torch.tensor(values, dtype=torch.float64, device=red_dev)
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(v) for v in t.tolist()]

    base = asdict(Map_Scenario())

    def timed(fn_warm, fn_block, repeat=True):
        """Warm-up, then blocks of the timed region (fn_block enqueues EXACTLY K steps), each bracketed by barrier +
        synchronize on both sides.  Returns (median over blocks of the max-over-ranks wall time [s], median HIP-event
        time of this rank's blocks [ms], number of blocks)."""
        fn_warm()

        def one():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            barrier()
            e0.record(stream)
            t0 = time.perf_counter()
            fn_block()
            e1.record(stream)
            while not e1.query():     # notice completion by polling, then the contract's synchronize (the work is complete: it
                pass                  # returns at once) - the rank's time is taken here, the MAX over ranks afterwards
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            barrier()
            return dt, e0.elapsed_time(e1)
        first = one()
        t_first = max_over_ranks([first[0]])[0]            # the same number of blocks on every rank
        reps = min(MAX_REPS, max(1, math.ceil(MIN_TIMED_S / max(t_first, 1e-7)))) if repeat else 1
        blocks = [first] + [one() for _ in range(reps - 1)]
        walls = max_over_ranks([b[0] for b in blocks])
        return float(np.median(walls)), float(np.median([b[1] for b in blocks])), reps

    # ---- workloads ------------------------------------------------------------------------------------------
    def run_rk4_single(N, steps, warmup, layout, variant, no_reuse=False):
        p = base | {"N": N}
        eq = LMAHeureuxPorosityDiff.from_scenario(p, device=local_rank)
        eq.use_stream(stream.cuda_stream)
        if variant >= 0:
            eq.set_option("rk4_variant", variant)
        if no_reuse:
            eq.set_option("no_reuse", 1)
        y0 = torch.from_numpy(synthetic(p, N)).cuda()
        buf = torch.zeros(eq.state_doubles(layout), dtype=torch.float64, device="cuda")
        eq.convert_layout_device(y0.data_ptr(), buf.data_ptr(), 0, layout)
        dt = 0.25 * (eq.Depths.length / N) ** 2
        # untimed settle phase (~60 ms of the same kernel): the chip's clocks ramp for tens of ms after idle, and a short
        # --steps run would otherwise measure the ramp; then the W warm-up steps, then the blocks of K timed steps
        # (as long as the timed call when that is longer: the streamed path is ONE launch per call, and a profile of this
        # command should see launches of one length)
        eq.integrate_rk4_device(buf.data_ptr(), dt, max(2000, steps), layout)
        wall, ev_ms, reps = timed(lambda: eq.integrate_rk4_device(buf.data_ptr(), dt, warmup, layout),
                                  lambda: eq.integrate_rk4_device(buf.data_ptr(), dt, steps, layout))
        eq.synchronize()   # (outside the timed region: surfaces a streamed run that gave up waiting - marl_synchronize)
        assert bool(torch.isfinite(buf).all()), "state went non-finite"
        eq.close()
        return wall, ev_ms, reps

    def run_sweep(N, B, steps, warmup, adaptive, variant=-1):
        # 16x16x16-style grid over the three knobs the reference's tests vary, B instances per GPU
        k = max(1, round(B ** (1 / 3)))
        idx = np.arange(B) + rank * B
        inst = [{"Phi0": float(0.5 + 0.3 * ((i % k) / max(k - 1, 1))),
                 "PhiIni": float(0.5 + 0.3 * (((i // k) % k) / max(k - 1, 1))),
                 "k3": float(10 ** (-2 + ((i // (k * k)) % k) / max(k - 1, 1))),
                 } for i in idx]
        for d in inst:
            d["PhiNR"] = d["PhiIni"]
            d["k4"] = d["k3"]
        p = base | {"N": N}
        eq = LMAHeureuxPorosityDiff.from_scenario(p, device=local_rank, instances=inst)
        eq.use_stream(stream.cuda_stream)
        if variant >= 0:
            eq.set_option("sweep_variant", variant)
        y0 = torch.from_numpy(np.stack([synthetic(p | d, N) for d in inst])).cuda()
        dx2 = (eq.Depths.length / N) ** 2
        stats = {}

        def go(n, buf):
            if adaptive:
                stats["res"] = eq.sweep_rk45_device(buf.data_ptr(), (0.0, 1.0e9), 0.5 * dx2, 1e-3, 1e-3, max_attempts=n)
            else:
                eq.sweep_rk4_device(buf.data_ptr(), 0.25 * dx2, n)
        warm, buf = y0.clone(), y0.clone()
        wall, ev_ms, reps = timed(lambda: go(warmup, warm), lambda: go(steps, buf), repeat=False)
        assert bool(torch.isfinite(buf).all()), "state went non-finite"
        info = {}
        if adaptive:
            acc = np.array([r.n_accepted for r in stats["res"]])
            rej = np.array([r.n_rejected for r in stats["res"]])
            assert np.all(acc + rej == steps), "every instance must have spent its attempt budget"
            info = {"accepted_steps_mean": float(acc.mean()), "rejected_steps_mean": float(rej.mean())}
        eq.close()
        return wall, ev_ms, reps, info

    def run_rk45_single(N, steps, warmup, layout, variant=-1):
        p = base | {"N": N}
        eq = LMAHeureuxPorosityDiff.from_scenario(p, device=local_rank)
        eq.use_stream(stream.cuda_stream)
        if variant >= 0:
            eq.set_option("rk45_variant", variant)
        y0 = torch.from_numpy(synthetic(p, N)).cuda()
        dx2 = (eq.Depths.length / N) ** 2
        out = {}

        def go(n, buf):
            out["res"] = eq.integrate_rk45_device(buf.data_ptr(), (0.0, 1.0e9), 0.5 * dx2, 1e-3, 1e-3, layout, max_attempts=n)
        bufs = []
        for _ in range(2):
            b = torch.zeros(eq.state_doubles(layout), dtype=torch.float64, device="cuda")
            eq.convert_layout_device(y0.data_ptr(), b.data_ptr(), 0, layout)
            bufs.append(b)
        wall, ev_ms, reps = timed(lambda: go(warmup, bufs[0]), lambda: go(steps, bufs[1]), repeat=False)
        r = out["res"]
        eq.close()
        return wall, ev_ms, reps, {"accepted_steps": r.n_accepted, "rejected_steps": r.n_rejected}

    def run_dd(N, steps, warmup):
        """BASELINE configs[4]: ONE grid of N cells split over all ranks, halo strips + step-control records in one
        all-gather per attempt (RCCL)."""
        from marlpde_amd.domain import DomainDecomposedRK45, owned_slice
        p = base | {"N": N}
        dd = DomainDecomposedRK45(p, N, device=local_rank)
        y0 = synthetic(p, N)
        dx2 = ((p["max_depth"] / p["Xstar"]) / N) ** 2
        own = owned_slice(y0, N, dd.begin, dd.end)
        out = {}

        ys = [torch.from_numpy(own.copy()).cuda() for _ in range(2)]   # inputs resident in HBM before the timed region starts

        def go(n, y):
            out["st"] = dd.integrate(y, (0.0, 1.0e9), 0.5 * dx2, 1e-3, 1e-3, max_attempts=n)
        wall, ev_ms, reps = timed(lambda: go(warmup, ys[0]), lambda: go(steps, ys[1]), repeat=False)
        st = out["st"]
        info = {"accepted_steps": int(st.n_accepted), "rejected_steps": int(st.n_rejected), "transport": dd.transport}
        dd.close()
        return wall, ev_ms, reps, info

    # ---- headline -------------------------------------------------------------------------------------------
    single = args.workload in ("rk4_single", "rk45_single", "dd_rk45")
    N = args.n or ((1 << 22) if args.workload == "dd_rk45" else (1 << 20) if single else 1024)
    defaults = {"rk4_single": (4000, 4000), "rk45_single": (2000, 20), "sweep_rk45": (2000, 20), "sweep_rk4": (2000, 20), "dd_rk45": (500, 16)}
    steps = args.steps if args.steps is not None else defaults[args.workload][0]
    warmup = args.warmup if args.warmup is not None else defaults[args.workload][1]
    extra = {}
    if args.workload == "dd_rk45":
        wall, ev_ms, reps, info = run_dd(N, steps, warmup)
        units = float(N) * steps / world       # per-rank share; `value` multiplies by world below
        workload = f"dd_rk45 ONE grid N={N} over {world} rank(s), halo exchange + all-gathered step control (BASELINE configs[4])"
        kernel, parallelism, scaling = "rk45_attempt_kernel", f"1-D domain decomposition over {world} rank(s); one all-gather (halo strips + step-control record) per attempted step", "strong"
        extra.update(info)
    elif args.workload == "rk4_single":
        wall, ev_ms, reps = run_rk4_single(N, steps, warmup, args.layout, args.variant, args.no_reuse)
        units = float(N) * steps
        workload = (f"rk4_fused_single_grid N={N} fp64 (north_star target size 2^20; BASELINE configs[1] is the same kernel at "
                    f"N=65536, reported under extra), dt=0.25dx^2" + (", transcendental reuse disabled" if args.no_reuse else ""))
        # marl_api.hip rk4_run: grids of >= 196 608 cells run the whole call as ONE dataflow launch (rk4_stream_kernel) when it
        # spans at least two fused levels (4 steps each above 262 144 cells, 8 below); the default warm-up equals the timed
        # call, so that every launch of the kernel in a profile of this command has the same length
        per = 16 if N <= 98304 else (8 if N <= 262144 else 4)
        streamed = args.variant < 0 and N >= 196608 and steps >= 2 * per
        kernel = "rk4_stream_kernel" if streamed else "rk4_fused_kernel"
        parallelism, scaling = f"{world} rank(s), each integrating its own grid; no collective in the data path", "weak"
    elif args.workload in ("sweep_rk45", "sweep_rk4"):
        wall, ev_ms, reps, info = run_sweep(N, args.batch, steps, warmup, args.workload == "sweep_rk45", args.variant)
        units = float(N) * args.batch * steps
        workload = (f"{args.workload} batch={args.batch} instances/GPU x N={N}, one workgroup per instance "
                    f"(BASELINE configs[2]/[3]); steps = attempted steps")
        kernel = "rk45_sweep_kernel" if args.workload == "sweep_rk45" else "rk4_sweep_kernel"
        parallelism, scaling = f"instances sharded over {world} rank(s); no collective in the data path", "weak"
        extra.update(info)
    else:
        wall, ev_ms, reps, info = run_rk45_single(N, steps, warmup, args.layout, args.variant)
        units = float(N) * steps
        workload = f"rk45_fused_single_grid N={N} fp64, rtol=atol=1e-3; steps = attempted steps"
        kernel, parallelism, scaling = "rk45_attempt_kernel", f"{world} rank(s), each integrating its own grid; no collective in the data path", "weak"
        extra.update(info)
    value = units * world / wall

    # ---- CPU baseline (rank 0 only; the oracle is the checker / baseline, never the product) -----------------
    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        from oracle import oracle as orc
        orc.build()
        ncores = min(len(os.sched_getaffinity(0)), 16)  # a 1-GPU box shares its host: 16 CPUs per GPU
        L = base["max_depth"] / base["Xstar"]
        if args.workload == "rk4_single":
            Nc, sc = N, (12 if N >= (1 << 19) else max(2, int(1.2e7 // N)))   # ~3 s single-thread + ~3 s all cores
            P = orc.params_from_dict(base | {"N": Nc})
            yc = synthetic(base | {"N": Nc}, Nc)
            dtc = 0.25 * (L / Nc) ** 2
            t0 = time.perf_counter(); orc.rk4(P, Nc, yc, dtc, sc); t1c = time.perf_counter() - t0
            os.environ.setdefault("OMP_NUM_THREADS", str(ncores))
            t0 = time.perf_counter(); orc.rk4(P, Nc, yc, dtc, 4 * sc, omp=True); tomp = time.perf_counter() - t0
            cpu = {"value": Nc * sc / t1c, "unit": "grid-point-steps/s", "cores": 1, "kind": "port",
                   "sample": f"oracle RK4 (C port of the reference's serial loop), N={Nc}, {sc} steps, 1 thread",
                   "value_all_cores": Nc * 4 * sc / tomp, "cores_all": ncores,
                   "sample_all_cores": f"same, OpenMP over cells, {4 * sc} steps, {ncores} threads"}
        elif single:
            Nc = min(N, 1 << 20)
            sc = max(3, int(1.2e7 // Nc))
            P = orc.params_from_dict(base | {"N": Nc})
            t0 = time.perf_counter()
            orc.rk45(P, Nc, synthetic(base | {"N": Nc}, Nc), 0.0, 1e9, 0.5 * (L / Nc) ** 2, 1e-3, 1e-3, max_attempts=sc, max_steps_out=1)
            t1c = time.perf_counter() - t0
            cpu = {"value": Nc * sc / t1c, "unit": "grid-point-steps/s", "cores": 1, "kind": "port",
                   "sample": f"oracle RK45 (C port, scipy-exact controller), N={Nc}, {sc} attempted steps, 1 thread"}
        else:
            Bc, sc = 16, 600
            inst = [{"Phi0": 0.5 + 0.3 * i / 15, "PhiIni": 0.5 + 0.3 * ((i * 7) % 16) / 15} for i in range(Bc)]
            t0 = time.perf_counter()
            for d in inst:
                d["PhiNR"] = d["PhiIni"]
                P = orc.params_from_dict(base | d | {"N": N})
                orc.rk45(P, N, synthetic(base | d | {"N": N}, N), 0.0, 1e9, 0.5 * (L / N) ** 2, 1e-3, 1e-3, max_attempts=sc,
                         max_steps_out=1)
            t1c = time.perf_counter() - t0
            cpu = {"value": N * Bc * sc / t1c, "unit": "grid-point-steps/s", "cores": 1, "kind": "port",
                   "sample": f"oracle RK45 (C port, scipy-exact controller), {Bc} instances x N={N} x {sc} attempts, 1 thread"}

    # ---- roofline of the dominant kernel ---------------------------------------------------------------------
    line = None
    if rank == 0:
        # achieved = ALGORITHMIC bytes per launch / average launch duration = 80 B x grid-point-steps / HIP-event time
        # of the timed block on the launch stream (launches are back to back, so gaps count against the kernel)
        achieved = BYTES_PER_POINT_STEP * units / (ev_ms * 1e-3) / 1e9  # GB/s
        # `bound` names what binds the kernel: the fused integrators advance several steps per pass over
        # the state, draw a fraction of the algorithmic bytes from HBM and are limited by fp64 VALU issue.  achieved / peak /
        # frac stay the task contract's figures - the EFFECTIVE rate of 80 algorithmic bytes per grid-point-step against the
        # HBM roofline (north_star's "% of HBM roofline"); valu_issue_frac is the fraction of the roofline that does bind.
        roof = {"bound": "fp64_valu", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": None, "kernel": kernel, "algorithmic_bytes_per_grid_point_step": BYTES_PER_POINT_STEP,
                "note": "achieved = algorithmic bytes (80 B x grid-point-steps of one launch) / launch duration (HIP events on the launch "
                        "stream): an effective rate against the one-step-per-pass HBM roofline, NOT the bandwidth the kernel draws "
                        "(hbm_measured_gbs: PMC bytes per launch / launch duration).  The kernel is bound by fp64 VALU issue: "
                        "valu_issue_frac = VALU wave-instructions of the launch (SQ_INSTS_VALU of the committed counter pass: a "
                        "property of code + input) / launch duration of THIS run / (1024 SIMDs x 2.4 GHz / 4 cycles)"}
        pmc = sqc = None
        for tag in ("r04", "r03", "r02", "r01"):
            f = os.path.join(ROOT, "profiles", f"{tag}_pmc_hbm_traffic.json")
            if pmc is None and os.path.exists(f):
                pmc = (tag, json.load(open(f)))
            f = os.path.join(ROOT, "profiles", f"{tag}_pmc_sq_counters.json")
            if sqc is None and os.path.exists(f):
                sqc = (tag, json.load(open(f)))
        # VALU issue fraction of the dominant kernel: instructions per grid-point-step from the counter pass of the same workload
        # and size (profiles/<tag>_pmc_sq_counters.json, "valu_insts_per_grid_point_step"), time from this run
        wl_key = {"rk4_single": "default" if N == (1 << 20) else ("n65536" if N == 65536 else None), "rk45_single": "rk45_single" if N == (1 << 20) else None,
                  "sweep_rk45": "sweep_rk45", "sweep_rk4": "sweep_rk4"}.get(args.workload)
        if args.no_reuse and wl_key == "default":
            wl_key = "default_no_reuse"
        if sqc and wl_key and args.variant < 0:
            ks = sqc[1].get("workloads", {}).get(wl_key, {})
            k = next((v for n, v in ks.items() if kernel in n and "valu_insts_per_grid_point_step" in v), None)
            if k:
                rate = k["valu_insts_per_grid_point_step"] * units / (ev_ms * 1e-3)
                roof["valu_issue_frac"] = rate / VALU_PEAK_WAVE_INSTS_PER_S
                roof["valu"] = {"wave_insts_per_grid_point_step": k["valu_insts_per_grid_point_step"], "wave_insts_per_s": rate,
                                "peak_wave_insts_per_s": VALU_PEAK_WAVE_INSTS_PER_S, "source": f"profiles/{sqc[0]}_pmc_sq_counters.json ({wl_key})"}
        if pmc and args.workload == "rk4_single" and N == (1 << 20) and args.variant < 0:
            want = "rk4_stream_kernel<256, 1, 4" if kernel == "rk4_stream_kernel" else "rk4_fused_kernel<256, 1, 1, 4"
            k = next((v for n, v in pmc[1]["kernels"].items() if want in n), None)
            if k:
                # one launch of this run = `steps` steps (streamed) or 4 steps (per-level launches); the PMC figure is per step
                per_step = k["hbm_bytes_per_launch"] / k.get("steps_per_launch", 4)
                roof["traffic"] = per_step * (steps if kernel == "rk4_stream_kernel" else 4)
                roof["hbm_measured_gbs"] = per_step * steps / (ev_ms * 1e-3) / 1e9
                roof["hbm_measured_frac_of_peak"] = roof["hbm_measured_gbs"] / HBM_PEAK_GBS
                roof["traffic_note"] = (f"HBM bytes per launch ({steps if kernel == 'rk4_stream_kernel' else 4} RK4 steps) from rocprofv3 --pmc FETCH_SIZE / "
                                        f"WRITE_SIZE, separate passes of launches of {k.get('steps_per_launch', 4)} steps scaled by the step count, "
                                        f"FETCH_SIZE doubled per the gfx950 calibration (profiles/{pmc[0]}_pmc_hbm_traffic.json, collected with "
                                        f"tools/profile_round.sh, not in this run); algorithmic bytes per step = {80 * N}")
        line = {
            "metric": "grid-point-steps/sec (5 fields, fp64)", "value": value, "unit": "grid-point-steps/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * wall / steps, "reps": reps,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload, "N": N, "instances_per_gpu": (1 if single else args.batch),
                       "layout": ("tiled" if args.layout else "field-major") if single else "field-major",
                       "parallelism": parallelism,
                       "timing": f"median of {reps} block(s) of {steps} steps, each bracketed by barrier + synchronize; max over ranks per block"},
            "roofline": roof, "cpu_baseline": cpu,
            # which build of the library produced this line (MARL_HIP_LIBRARY / MARL_HIP_OPTIONS are kernel-lab overrides: a line
            # taken with either set says so)
            "library": dict(zip(("path", "sha256_16"), _abi.library_fingerprint())) | {
                "overridden": bool(os.environ.get("MARL_HIP_LIBRARY")), "lab_options": os.environ.get("MARL_HIP_OPTIONS", "")},
        }

    # ---- extras: the other BASELINE configs, best effort under a deadline -----------------------------------
    printed = threading.Lock()

    def emit(timed_out=False):
        if not printed.acquire(blocking=False):
            return
        if rank == 0:
            line["extras_timed_out"] = bool(timed_out)   # top level: a hung extra must not read as a clean run
            if timed_out:
                extra["extras_timed_out_after_s"] = EXTRAS_DEADLINE_S
            if extra:
                line["extra"] = extra
            os.write(json_fd, (json.dumps(line) + "\n").encode())

    def deadline():
        emit(timed_out=True)
        # a hung collective cannot be unwound: leave.  The headline line is out and valid (it was measured before the extras
        # started), so the exit code stays 0 unless MARL_BENCH_TIMEOUT_RC asks otherwise (the rehearsal test sets it); the
        # line itself says "extras_timed_out": true at top level
        # ... with more than one rank the default is non-zero: there the likeliest cause is a collective that never completes (a broken
        # RCCL set-up on the multi-GPU node), and that must be red, not a flag inside a green line
        os._exit(int(os.environ.get("MARL_BENCH_TIMEOUT_RC", "3" if world > 1 else "0")))

    do_extras = args.workload == "rk4_single" and args.n is None and args.variant < 0 and not args.no_extras and not args.no_reuse
    if do_extras:
        dog = threading.Timer(EXTRAS_DEADLINE_S, deadline)
        dog.daemon = True
        dog.start()

        def guarded(name, fn):
            try:
                barrier()
                fn()
            except Exception as e:   # noqa: BLE001 - an extra must never cost the headline
                extra[name + "_error"] = f"{type(e).__name__}: {e}"

        # MARL_BENCH_LIGHT_EXTRAS=1 (rehearsals of many ranks on ONE device, tests/test_gpu_bench.py): the same code paths at reduced sizes
        light = os.environ.get("MARL_BENCH_LIGHT_EXTRAS") == "1"

        def x_n65536():
            w, _, r = run_rk4_single(65536, 4000, 200, args.layout, -1)
            v = 65536.0 * 4000 * world / w
            extra["BASELINE_configs1_rk4_N65536"] = {"value": v, "unit": "grid-point-steps/s", "n_ranks": world, "reps": r,
                                                     "effective_frac_of_hbm_roofline_per_gpu": BYTES_PER_POINT_STEP * v / world / 1e9 / HBM_PEAK_GBS}

        def x_no_reuse():
            w, _, r = run_rk4_single(1 << 20, steps, warmup, args.layout, -1, no_reuse=True)
            v = float(1 << 20) * steps * world / w
            if rank == 0:   # beside the headline's fraction, on the line the driver parses: the input dependence of the headline
                line["roofline"]["frac_reuse_off"] = BYTES_PER_POINT_STEP * v / world / 1e9 / HBM_PEAK_GBS
                line["roofline"]["frac_reuse_off_note"] = ("same kernel with transcendental reuse forced off (extra.rk4_N1048576_no_reuse; wall-clock based): "
                                                           "the input-independent floor; `frac` needs a state smooth enough for the stage-to-stage expansions")
            extra["rk4_N1048576_no_reuse"] = {"value": v, "unit": "grid-point-steps/s", "n_ranks": world,
                                              "effective_frac_of_hbm_roofline_per_gpu": BYTES_PER_POINT_STEP * v / world / 1e9 / HBM_PEAK_GBS,
                                              "note": "same kernel, no evaluation may reuse the transcendentals of an earlier one (what a rough "
                                                      "state does wave by wave): the input-independent floor of the headline"}

        def x_sweep():
            B, att = (256, 200) if light else (4096, 2000)
            w, _, _, info = run_sweep(1024, B, att, 20, True)
            extra["BASELINE_configs2_3_sweep_rk45"] = {"value": 1024.0 * B * att * world / w, "unit": "grid-point-steps/s (attempted steps)",
                                                       "n_ranks": world, "instances_total": B * world, "N": 1024, "attempts": att,
                                                       "parallelism": "instances sharded evenly over the ranks; no collective", **info}

        def x_dd():
            Nd, att = (1 << 20, 60) if light else (1 << 22, 500)
            w, _, _, info = run_dd(Nd, att, 16)
            extra["BASELINE_configs4_dd_rk45"] = {"value": float(Nd) * att / w, "unit": "grid-point-steps/s (attempted steps)",
                                                  "n_ranks": world, "N": Nd, "attempts": att, "scaling": "strong",
                                                  "parallelism": "1-D domain decomposition; one all-gather (halo strips + record) per attempt", **info}
        def x_radau():
            # the reference's DEFAULT solver (implicit Radau) through marl_integrate_radau: time to solution of
            # Scenario A to T* = 13 190 yr (rtol = atol = 1e-3), every rank its own copy; the CPU oracle beside it on rank 0
            res = {}
            for Nr in (200, 4000, 64000):
                pr = base | {"Phi0": 0.6, "PhiIni": 0.5, "PhiNR": 0.6, "N": Nr}
                y0 = np.concatenate([np.full(Nr, pr[k]) for k in ("CAIni", "CCIni", "cCaIni", "cCO3Ini", "PhiIni")])
                eq = LMAHeureuxPorosityDiff.from_scenario(pr, device=local_rank)
                eq.use_stream(stream.cuda_stream)
                eq.integrate_radau(y0, (0.0, 1e-4), 1e-6, 1e-3, 1e-3, events=False)     # allocations, module load
                t0 = time.perf_counter()
                r = eq.integrate_radau(y0, (0.0, 1.0), 1e-6, 1e-3, 1e-3)
                e = {"seconds": time.perf_counter() - t0, "status": r.status, "steps": r.n_accepted, "nfev": r.nfev, "njev": r.njev, "nlu": r.nlu}
                eq.close()
                if rank == 0 and Nr <= 4000 and not args.no_cpu_baseline:
                    from oracle import oracle as orc
                    t0 = time.perf_counter()
                    _, st, *_ = orc.radau(orc.params_from_dict(pr), Nr, y0, 0.0, 1.0, 1e-6, 1e-3, 1e-3)
                    e["cpu_oracle_seconds_1_core"] = time.perf_counter() - t0
                    e["cpu_oracle_nfev_njev_nlu"] = [int(st.nfev), int(st.njev), int(st.nlu)]
                res[f"N{Nr}"] = e
            extra["implicit_radau_scenarioA_to_Tstar"] = res
            # the other implicit method of the reference's Solver, scipy BDF semantics (marl_integrate_bdf), same scenario and sizes
            resb = {}
            for Nr in (200, 4000, 64000):
                pr = base | {"Phi0": 0.6, "PhiIni": 0.5, "PhiNR": 0.6, "N": Nr}
                y0 = np.concatenate([np.full(Nr, pr[k]) for k in ("CAIni", "CCIni", "cCaIni", "cCO3Ini", "PhiIni")])
                eq = LMAHeureuxPorosityDiff.from_scenario(pr, device=local_rank)
                eq.use_stream(stream.cuda_stream)
                eq.integrate_bdf(y0, (0.0, 1e-4), 1e-6, 1e-3, 1e-3, events=False)
                t0 = time.perf_counter()
                r = eq.integrate_bdf(y0, (0.0, 1.0), 1e-6, 1e-3, 1e-3)
                e = {"seconds": time.perf_counter() - t0, "status": r.status, "steps": r.n_accepted, "nfev": r.nfev, "njev": r.njev, "nlu": r.nlu}
                eq.close()
                if rank == 0 and Nr <= 4000 and not args.no_cpu_baseline:
                    from oracle import oracle as orc
                    t0 = time.perf_counter()
                    _, st, *_ = orc.bdf(orc.params_from_dict(pr), Nr, y0, 0.0, 1.0, 1e-6, 1e-3, 1e-3)
                    e["cpu_oracle_seconds_1_core"] = time.perf_counter() - t0
                    e["cpu_oracle_nfev_njev_nlu"] = [int(st.nfev), int(st.njev), int(st.nlu)]
                resb[f"N{Nr}"] = e
            resb["note"] = ("status -1 on the finer grids is the METHOD: scipy's BDF itself stops at t = 0.0853 T* with 'Required step size is less than spacing "
                            "between numbers' at N = 4000 (rtol = atol = 1e-3); the oracle and the GPU path reproduce it (DESIGN.md 8)")
            extra["implicit_bdf_scenarioA_to_Tstar"] = resb
            # a SWEEP with the same solver: 512 scenarios (Phi0 x PhiIni x k3 = k4 grid), N = 200, all advanced together
            Bs, Ns = 512, 200
            kk = 8
            inst = [{"Phi0": 0.5 + 0.2 * ((i % kk) / (kk - 1)), "PhiIni": 0.5 + 0.2 * (((i // kk) % kk) / (kk - 1)),
                     "k3": 10 ** (-2 + ((i // (kk * kk)) % kk) / (kk - 1))} for i in range(Bs)]
            for d in inst:
                d["PhiNR"] = d["PhiIni"]
                d["k4"] = d["k3"]
            ps = base | {"N": Ns}
            y0 = np.stack([np.concatenate([np.full(Ns, (ps | d)[q]) for q in ("CAIni", "CCIni", "cCaIni", "cCO3Ini", "PhiIni")]) for d in inst])
            eq = LMAHeureuxPorosityDiff.from_scenario(ps, device=local_rank, instances=inst)
            eq.use_stream(stream.cuda_stream)
            yd = torch.from_numpy(y0).cuda()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rs = eq.sweep_radau_device(yd.data_ptr(), (0.0, 1.0), 1e-6, 1e-3, 1e-3)
            sw = {"instances": Bs, "N": Ns, "seconds": time.perf_counter() - t0, "reached_Tstar": int(sum(r.status == 0 for r in rs)),
                  "nfev_median": float(np.median([r.nfev for r in rs])), "nfev_max": int(max(r.nfev for r in rs))}
            # the same sweep WITH the monitors' root times located per instance (what the reference stores for every run)
            yd = torch.from_numpy(y0).cuda()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rs = eq.sweep_radau_device(yd.data_ptr(), (0.0, 1.0), 1e-6, 1e-3, 1e-3, events=True, max_events=2048)
            sw["seconds_with_event_roots"] = time.perf_counter() - t0
            sw["event_roots_located"] = int(sum(len(t) for r in rs for t in r.t_events))
            eq.close()
            if rank == 0 and not args.no_cpu_baseline:
                from oracle import oracle as orc
                sample = list(range(0, Bs, Bs // 16))
                t0 = time.perf_counter()
                for b in sample:
                    orc.radau(orc.params_from_dict(ps | inst[b]), Ns, y0[b], 0.0, 1.0, 1e-6, 1e-3, 1e-3)
                sw["cpu_oracle_seconds_1_core_extrapolated"] = (time.perf_counter() - t0) / len(sample) * Bs
            extra["implicit_radau_sweep"] = sw

        guarded("BASELINE_configs1_rk4_N65536", x_n65536)
        guarded("rk4_N1048576_no_reuse", x_no_reuse)
        guarded("BASELINE_configs2_3_sweep_rk45", x_sweep)
        guarded("BASELINE_configs4_dd_rk45", x_dd)
        if not light:
            guarded("implicit_radau_scenarioA_to_Tstar", x_radau)
        dog.cancel()
    emit()
    if use_dist:
        try:
            dist.destroy_process_group()
        except Exception:   # noqa: BLE001
            pass


if __name__ == "__main__":
    main()
