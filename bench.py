#!/usr/bin/env python3
"""Throughput of the hot path on MI355X: grid-point-steps/s of the fused five-field RK integrators.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload rk4_single|sweep_rk45|rk45_single] ...

Default (N=1): BASELINE.json's headline - ONE grid of 2^20 depth cells, fixed-step classical RK4, fp64,
fused HIP stencil (configs[1] at the size the north_star quotes its target on).  A "step" is one RK4
step of the whole grid.  With --gpus N each rank integrates its own independent grid of the same size
(weak scaling, no data-path collective - a sweep over independent high-resolution columns).
`--workload sweep_rk45` is BASELINE config 3/4: 4096 instances x 1024 cells per GPU, adaptive RK45 with
per-instance controllers, a "step" being one attempted step of every instance.

One JSON line is printed by rank 0 (contract in the task description); `roofline` is for the dominant
kernel with HIP-event timing on the launch stream, `cpu_baseline` is the oracle (a C port of the
reference's serial loop) timed on this host on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
BYTES_PER_POINT_STEP = 80.0    # 5 fields x 8 B read + 5 x 8 B written per grid-point-step


def synthetic(p, N, batch=None):
    """SURVEY.md 8d synthetic input: initial values x (1 + 0.01 sin(2 pi 8 x / L)); no RNG."""
    L = p["max_depth"] / p["Xstar"]
    x = (np.arange(N) + 0.5) * (L / N)
    y = np.stack([np.full(N, p[k]) for k in ("CAIni", "CCIni", "cCaIni", "cCO3Ini", "PhiIni")])
    return (y * (1.0 + 0.01 * np.sin(2 * np.pi * 8 * x / L))).ravel()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="rk4_single", choices=["rk4_single", "sweep_rk45", "sweep_rk4", "rk45_single", "dd_rk45"])
    ap.add_argument("--n", type=int, default=None, help="cells per grid (default 2^20 single, 1024 sweep)")
    ap.add_argument("--batch", type=int, default=4096, help="instances per GPU for the sweep workloads")
    ap.add_argument("--layout", type=int, default=1, help="device layout of single-grid runs: 0 field-major, 1 tiled")
    ap.add_argument("--variant", type=int, default=-1, help="kernel variant (see DESIGN.md); -1 = default")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--extras", action="store_true", help="also time BASELINE configs 2 and 3 and add them under `extra`")
    args = ap.parse_args()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool (RCCL / tensor sharing)
    import torch
    import torch.distributed as dist
    from dataclasses import asdict
    from marlpde_amd.LHeureux_model import LMAHeureuxPorosityDiff
    from marlpde_amd.parameters import Map_Scenario

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the package has no CPU path)"
    # rehearsal of the N > 1 path on a one-GPU box: MARL_BENCH_BACKEND=gloo MARL_BENCH_ONE_DEVICE=1 (all ranks on cuda:0)
    if os.environ.get("MARL_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        backend = os.environ.get("MARL_BENCH_BACKEND", "nccl")   # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    stream = torch.cuda.current_stream()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    base = asdict(Map_Scenario())
    single = args.workload in ("rk4_single", "rk45_single", "dd_rk45")
    N = args.n or ((1 << 20) if single else 1024)
    steps = args.steps if args.steps is not None else (4000 if single else 2000)
    warmup = args.warmup if args.warmup is not None else (200 if single else 20)   # ~6 ms: lets the clocks settle

    def timed(fn_warm, fn_timed):
        """warm-up, then the timed region bracketed by barrier + synchronize; returns (wall s, event ms)."""
        fn_warm()
        barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(stream)
        fn_timed()
        e1.record(stream)
        barrier()
        wall = time.perf_counter() - t0
        return wall, e0.elapsed_time(e1)

    def run_rk4_single(N, steps, warmup, layout, variant):
        p = base | {"N": N}
        eq = LMAHeureuxPorosityDiff.from_scenario(p, device=local_rank)
        eq.use_stream(stream.cuda_stream)
        if variant >= 0:
            eq.set_option("rk4_variant", variant)
        y0 = torch.from_numpy(synthetic(p, N)).cuda()
        buf = torch.zeros(eq.state_doubles(layout), dtype=torch.float64, device="cuda")
        eq.convert_layout_device(y0.data_ptr(), buf.data_ptr(), 0, layout)
        dt = 0.25 * (eq.Depths.length / N) ** 2
        # untimed settle phase (~60 ms of the same kernel): the chip's clocks ramp for tens of ms after idle, and a short
        # --steps run would otherwise measure the ramp; then the W warm-up steps, then the K timed steps
        eq.integrate_rk4_device(buf.data_ptr(), dt, 2000, layout)
        wall, ev_ms = timed(lambda: eq.integrate_rk4_device(buf.data_ptr(), dt, warmup, layout),
                            lambda: eq.integrate_rk4_device(buf.data_ptr(), dt, steps, layout))
        assert bool(torch.isfinite(buf).all()), "state went non-finite"
        eq.close()
        return wall, ev_ms, dt

    def run_sweep(N, B, steps, warmup, adaptive):
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
        if args.variant >= 0:
            eq.set_option("sweep_variant", args.variant)
        y0 = torch.from_numpy(np.stack([synthetic(p | d, N) for d in inst])).cuda()
        dx2 = (eq.Depths.length / N) ** 2
        stats = {}

        def go(n, buf):
            if adaptive:
                stats["res"] = eq.sweep_rk45_device(buf.data_ptr(), (0.0, 1.0e9), 0.5 * dx2, 1e-3, 1e-3, max_attempts=n)
            else:
                eq.sweep_rk4_device(buf.data_ptr(), 0.25 * dx2, n)
        warm, buf = y0.clone(), y0.clone()
        wall, ev_ms = timed(lambda: go(warmup, warm), lambda: go(steps, buf))
        assert bool(torch.isfinite(buf).all()), "state went non-finite"
        info = {}
        if adaptive:
            acc = np.array([r.n_accepted for r in stats["res"]])
            rej = np.array([r.n_rejected for r in stats["res"]])
            assert np.all(acc + rej == steps), "every instance must have spent its attempt budget"
            info = {"accepted_steps_mean": float(acc.mean()), "rejected_steps_mean": float(rej.mean())}
        eq.close()
        return wall, ev_ms, info

    def run_rk45_single(N, steps, warmup, layout):
        p = base | {"N": N}
        eq = LMAHeureuxPorosityDiff.from_scenario(p, device=local_rank)
        eq.use_stream(stream.cuda_stream)
        if args.variant >= 0:
            eq.set_option("rk45_variant", args.variant)
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
        wall, ev_ms = timed(lambda: go(warmup, bufs[0]), lambda: go(steps, bufs[1]))
        r = out["res"]
        eq.close()
        return wall, ev_ms, {"accepted_steps": r.n_accepted, "rejected_steps": r.n_rejected}

    def run_dd(N, steps, warmup):
        """BASELINE config 5: ONE grid of N cells split over all ranks, RCCL halo exchange + all-gathered control."""
        from marlpde_amd.domain import DomainDecomposedRK45, owned_slice
        p = base | {"N": N}
        dd = DomainDecomposedRK45(p, N, device=local_rank)
        y0 = synthetic(p, N)
        dx2 = ((p["max_depth"] / p["Xstar"]) / N) ** 2
        own = owned_slice(y0, N, dd.begin, dd.end)
        out = {}

        def go(n):
            y = torch.from_numpy(own.copy()).cuda()
            out["st"] = dd.integrate(y, (0.0, 1.0e9), 0.5 * dx2, 1e-3, 1e-3, max_attempts=n)
        wall, ev_ms = timed(lambda: go(warmup), lambda: go(steps))
        st = out["st"]
        dd.close()
        return wall, ev_ms, {"accepted_steps": int(st.n_accepted), "rejected_steps": int(st.n_rejected)}

    extra = {}
    if args.workload == "dd_rk45":
        N = args.n or (1 << 22)
        steps = args.steps if args.steps is not None else 500
        warmup = args.warmup if args.warmup is not None else 16
        wall, ev_ms, info = run_dd(N, steps, warmup)
        units = float(N) * steps / world       # per-rank share; `value` multiplies by world below
        workload = f"dd_rk45 ONE grid N={N} over {world} rank(s), halo exchange + all-gathered step control (BASELINE configs[4])"
        kernel = "rk45_attempt_kernel"
        extra.update(info)
    elif args.workload == "rk4_single":
        wall, ev_ms, dt = run_rk4_single(N, steps, warmup, args.layout, args.variant)
        units = float(N) * steps
        workload = f"rk4_fused_single_grid N={N} fp64 (north_star target size 2^20; BASELINE configs[1] is the same kernel at N=65536, reported under extra), dt=0.25dx^2"
        kernel = "rk4_fused_kernel"
    elif args.workload in ("sweep_rk45", "sweep_rk4"):
        wall, ev_ms, info = run_sweep(N, args.batch, steps, warmup, args.workload == "sweep_rk45")
        units = float(N) * args.batch * steps
        workload = (f"{args.workload} batch={args.batch} instances/GPU x N={N}, one workgroup per instance "
                    f"(BASELINE configs[2]/[3]); steps = attempted steps")
        kernel = "rk45_sweep_kernel" if args.workload == "sweep_rk45" else "rk4_sweep_kernel"
        extra.update(info)
    else:
        wall, ev_ms, info = run_rk45_single(N, steps, warmup, args.layout)
        units = float(N) * steps
        workload = f"rk45_fused_single_grid N={N} fp64, rtol=atol=1e-3; steps = attempted steps"
        kernel = "rk45_attempt_kernel"
        extra.update(info)

    # max over ranks of the wall time of the timed region
    wall_t = torch.tensor([wall], dtype=torch.float64, device="cuda" if (world == 1 or dist.get_backend() == "nccl") else "cpu")
    if world > 1:
        dist.all_reduce(wall_t, op=dist.ReduceOp.MAX)
    wall_max = float(wall_t.item())
    value = units * world / wall_max

    # BASELINE.json configs[1] (N = 65 536, fused RK4) next to the north_star headline size: 4000 steps, ~15 ms
    if rank == 0 and world == 1 and args.workload == "rk4_single" and args.n is None and args.variant < 0:
        w2, _, _ = run_rk4_single(65536, 4000, 200, args.layout, -1)
        extra["BASELINE_configs1_rk4_N65536"] = {"value": 65536.0 * 4000 / w2, "unit": "grid-point-steps/s",
                                                 "frac_of_hbm_roofline": BYTES_PER_POINT_STEP * 65536.0 * 4000 / w2 / 1e9 / HBM_PEAK_GBS}
    if rank == 0 and args.extras and world == 1:
        w2, e2, _ = run_rk4_single(65536, 10000, 16, args.layout, -1)
        extra["config2_rk4_N65536_gps"] = 65536.0 * 10000 / w2
        w3, e3, info3 = run_sweep(1024, 4096, 2000, 20, True)
        extra["config3_sweep_rk45_4096x1024_gps"] = 1024.0 * 4096 * 2000 / w3
        extra["config3_info"] = info3

    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        from oracle import oracle as orc
        orc.build()
        ncores = min(len(os.sched_getaffinity(0)), 16)  # a 1-GPU box shares its host: 16 CPUs per GPU
        L = base["max_depth"] / base["Xstar"]
        if args.workload == "rk4_single":
            Nc, sc = N, (2 if N >= (1 << 19) else max(2, int(2e6 // N)))
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
            sc = max(3, int(4e6 // Nc))
            P = orc.params_from_dict(base | {"N": Nc})
            t0 = time.perf_counter()
            orc.rk45(P, Nc, synthetic(base | {"N": Nc}, Nc), 0.0, 1e9, 0.5 * (L / Nc) ** 2, 1e-3, 1e-3, max_attempts=sc, max_steps_out=1)
            t1c = time.perf_counter() - t0
            cpu = {"value": Nc * sc / t1c, "unit": "grid-point-steps/s", "cores": 1, "kind": "port",
                   "sample": f"oracle RK45 (C port, scipy-exact controller), N={Nc}, {sc} attempted steps, 1 thread"}
        else:
            Bc, sc = 16, 200
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

    if rank == 0:
        # dominant kernel: algorithmic bytes per launch / average launch duration (HIP events around the timed
        # region on the launch stream; launches are back to back, so gaps count against the kernel)
        per_launch = {"rk4_single": 4 if (N > 262144 and args.variant < 0) else None}.get(args.workload)
        achieved = BYTES_PER_POINT_STEP * units / (ev_ms * 1e-3) / 1e9  # GB/s
        traffic, traffic_note = None, "no PMC summary committed for this kernel variant"
        pmc_file = os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic.json")
        if per_launch and args.workload == "rk4_single" and N == (1 << 20) and os.path.exists(pmc_file):
            k = next((v for n, v in json.load(open(pmc_file))["kernels"].items() if "rk4_fused_kernel<256, 1, 1, 4" in n), None)
            if k:
                traffic = k["hbm_bytes_per_launch"]
                traffic_note = ("HBM bytes per launch (4 RK4 steps) from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, "
                                "FETCH_SIZE doubled per the gfx950 calibration (profiles/r01_pmc_hbm_traffic.json); "
                                f"algorithmic bytes per launch = {k['algorithmic_bytes_per_launch']}")
        line = {
            "metric": "grid-point-steps/sec (5 fields, fp64)", "value": value, "unit": "grid-point-steps/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * wall_max / steps,
            "higher_is_better": True, "scaling": "strong" if args.workload == "dd_rk45" else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload, "N": N, "instances_per_gpu": (1 if single else args.batch),
                       "layout": ("tiled" if args.layout else "field-major") if single else "field-major",
                       "parallelism": f"{world} independent rank(s), no collectives in the data path"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note, "kernel": kernel,
                         "algorithmic_bytes_per_grid_point_step": BYTES_PER_POINT_STEP,
                         "note": "achieved = 80 B x grid-point-steps / HIP-event time; the kernel is fp64-VALU-bound, see DESIGN.md"},
            "cpu_baseline": cpu,
        }
        if extra:
            line["extra"] = extra
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
