#!/usr/bin/env python3
"""A sweep of scenarios with the reference's default solver: B instances (Phi0 x PhiIni x k3 = k4 grid, the knobs the reference's tests
turn), N = 200, t 0 -> T*, Radau rtol = atol = 1e-3 - all instances advanced together on one MI355X (marl_sweep_radau_dev), against one
core of the host running the CPU oracle instance by instance (a sample, extrapolated).   python tools/radau_sweep_timing.py [B ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch  # noqa: E402
from common import scenario  # noqa: E402
from marlpde_amd.LHeureux_model import LMAHeureuxPorosityDiff  # noqa: E402
from oracle import oracle as orc  # noqa: E402

orc.build()
N = 200
base = scenario("default", N)
for B in [int(a) for a in sys.argv[1:]] or [64, 512, 4096]:
    k = max(1, round(B ** (1 / 3)))
    inst = []
    for i in range(B):
        d = {"Phi0": 0.5 + 0.2 * ((i % k) / max(k - 1, 1)), "PhiIni": 0.5 + 0.2 * (((i // k) % k) / max(k - 1, 1)),
             "k3": 10 ** (-2 + ((i // (k * k)) % k) / max(k - 1, 1))}
        d["PhiNR"] = d["PhiIni"]
        d["k4"] = d["k3"]
        inst.append(d)
    y0 = np.stack([np.concatenate([np.full(N, (base | d)[q]) for q in ("CAIni", "CCIni", "cCaIni", "cCO3Ini", "PhiIni")]) for d in inst])
    eq = LMAHeureuxPorosityDiff.from_scenario(base, device=0, instances=inst)
    yd = torch.from_numpy(y0).cuda()
    torch.cuda.synchronize()
    t0 = time.time()
    res = eq.sweep_radau_device(yd.data_ptr(), (0.0, 1.0), 1e-6, 1e-3, 1e-3)
    tg = time.time() - t0
    eq.close()
    nfev = np.array([r.nfev for r in res])
    ok = sum(r.status == 0 for r in res)
    sample = list(range(0, B, max(1, B // 16)))[:16]
    t0 = time.time()
    for b in sample:
        orc.radau(orc.params_from_dict(base | inst[b]), N, y0[b], 0.0, 1.0, 1e-6, 1e-3, 1e-3)
    tc = (time.time() - t0) / len(sample) * B
    print(f"B={B:5d} instances x N={N}: GPU sweep {tg:7.3f} s ({ok} reached T*; nfev min/median/max {nfev.min()}/{int(np.median(nfev))}/{nfev.max()}) | "
          f"CPU oracle, 1 core, instance by instance ~{tc:7.1f} s (from {len(sample)} samples) | ratio {tc / tg:6.1f}x", flush=True)
