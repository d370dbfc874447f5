#!/usr/bin/env python3
"""The implicit (Radau / BDF) path as a profiling target:  rocprofv3 --kernel-trace --stats -- python3 tools/radau_profile.py [what ...]
what: single (Scenario A to T* at N = 200, 16 000, 64 000; MARL_RADAU_NS=64000 picks other sizes), bdf (the same at N = 200 with BDF),
sweep (512 scenarios, N = 200).
Prints wall times; the kernel statistics / counters come from the profiler around it (tools/profile_round.sh)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch  # noqa: E402
from common import scenario  # noqa: E402
from marlpde_amd.LHeureux_model import LMAHeureuxPorosityDiff  # noqa: E402

what = sys.argv[1:] or ["single", "bdf", "sweep"]


def y0_of(p, N):
    return np.concatenate([np.full(N, p[k]) for k in ("CAIni", "CCIni", "cCaIni", "cCO3Ini", "PhiIni")])


if "single" in what or "bdf" in what:
    for method in [m for m in ("single", "bdf") if m in what]:
        sizes = [int(x) for x in os.environ["MARL_RADAU_NS"].split(",")] if os.environ.get("MARL_RADAU_NS") else None
        for N in sizes or ((200, 16000, 64000) if method == "single" else (200,)):
            p = scenario("A", N)
            eq = LMAHeureuxPorosityDiff.from_scenario(p, device=0)
            run = eq.integrate_radau if method == "single" else eq.integrate_bdf
            run(y0_of(p, N), (0.0, 1e-4), 1e-6, 1e-3, 1e-3, events=False)
            t0 = time.time()
            r = run(y0_of(p, N), (0.0, 1.0), 1e-6, 1e-3, 1e-3)
            print(f"{'radau' if method == 'single' else 'bdf'} N={N}: {time.time() - t0:.4f} s, nfev {r.nfev} njev {r.njev} nlu {r.nlu} steps {r.n_accepted} status {r.status}", flush=True)
            eq.close()
if "sweep" in what:
    N, B = 200, int(os.environ.get("MARL_RADAU_SWEEP_B", "512"))
    k = max(1, round(B ** (1 / 3)))
    base = scenario("default", N)
    inst = []
    for i in range(B):
        d = {"Phi0": 0.5 + 0.2 * ((i % k) / (k - 1)), "PhiIni": 0.5 + 0.2 * (((i // k) % k) / (k - 1)), "k3": 10 ** (-2 + ((i // (k * k)) % k) / (k - 1))}
        d["PhiNR"], d["k4"] = d["PhiIni"], d["k3"]
        inst.append(d)
    y0 = np.stack([y0_of(base | d, N) for d in inst])
    eq = LMAHeureuxPorosityDiff.from_scenario(base, device=0, instances=inst)
    yd = torch.from_numpy(y0).cuda()
    torch.cuda.synchronize()
    t0 = time.time()
    res = eq.sweep_radau_device(yd.data_ptr(), (0.0, 1.0), 1e-6, 1e-3, 1e-3)
    print(f"radau sweep B={B} N={N}: {time.time() - t0:.3f} s, {sum(r.status == 0 for r in res)} reached T*, nfev max {max(r.nfev for r in res)}", flush=True)
    eq.close()
