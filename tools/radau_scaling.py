#!/usr/bin/env python3
"""Implicit path at growing resolution: Scenario A to T* = 13 190 yr with Radau (rtol = atol = 1e-3), MI355X (marl_integrate_radau)
against the CPU oracle (banded LU, one core of the same host).  Explicit RK45 needs ~0.9 dx^2 steps (3e5 steps at N = 200,
3e9 at N = 20 000); Radau's step count does not depend on N.      python tools/radau_scaling.py [N ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from common import scenario  # noqa: E402
from marlpde_amd.LHeureux_model import LMAHeureuxPorosityDiff  # noqa: E402
from oracle import oracle as orc  # noqa: E402

orc.build()
for N in [int(a) for a in sys.argv[1:]] or [200, 1000, 4000, 16000, 64000]:
    p = scenario("A", N)
    y0 = np.concatenate([np.full(N, p[k]) for k in ("CAIni", "CCIni", "cCaIni", "cCO3Ini", "PhiIni")])
    eq = LMAHeureuxPorosityDiff.from_scenario(p, device=0)
    eq.integrate_radau(y0, (0.0, 1e-4), 1e-6, 1e-3, 1e-3, events=False)      # warm-up (module load, allocations)
    t0 = time.time()
    r = eq.integrate_radau(y0, (0.0, 1.0), 1e-6, 1e-3, 1e-3)
    tg = time.time() - t0
    line = f"N={N:6d}  GPU {tg:7.3f} s  status {r.status} steps {r.n_accepted} nfev {r.nfev} njev {r.njev} nlu {r.nlu}"
    if N <= 16000:
        t0 = time.time()
        y, st, *_ = orc.radau(orc.params_from_dict(p), N, y0, 0.0, 1.0, 1e-6, 1e-3, 1e-3)
        tc = time.time() - t0
        line += (f" | oracle (1 core) {tc:7.3f} s steps {st.n_accepted} nfev {st.nfev} njev {st.njev} nlu {st.nlu}"
                 f" | max |GPU - oracle| {np.max(np.abs(r.y_final - y)):.2e}")
    print(line, flush=True)
    eq.close()
