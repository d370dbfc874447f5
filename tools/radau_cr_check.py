#!/usr/bin/env python3
"""Cyclic reduction in front of PCR (csrc/marl_radau_cr.h) against plain PCR on ONE box: Scenario A to T* with Radau and BDF
(rtol = atol = 1e-3), time, scipy-style statistics, largest difference of the final states - and the CPU oracle (banded LU) where it
finishes in seconds.        python tools/radau_cr_check.py [N ...]"""
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
ORACLE_MAX_N = int(os.environ.get("MARL_CR_ORACLE_MAX_N", "4000"))
for N in [int(a) for a in sys.argv[1:]] or [200, 1000, 4000, 16000, 64000]:
    p = scenario("A", N)
    y0 = np.concatenate([np.full(N, p[k]) for k in ("CAIni", "CCIni", "cCaIni", "cCO3Ini", "PhiIni")])
    for method in ("radau", "bdf"):
        res = {}
        for label, cr in (("pcr", 0), ("cr+pcr", -1 if N >= 2048 else 3)):
            eq = LMAHeureuxPorosityDiff.from_scenario(p, device=0)
            eq.set_option("radau_cr", cr)
            run = eq.integrate_radau if method == "radau" else eq.integrate_bdf
            run(y0, (0.0, 1e-4), 1e-6, 1e-3, 1e-3, events=False)      # warm-up (module load, allocations)
            best = None
            for _ in range(2):
                t0 = time.time()
                r = run(y0, (0.0, 1.0), 1e-6, 1e-3, 1e-3)
                dt = time.time() - t0
                best = dt if best is None else min(best, dt)
            res[label] = r
            print(f"{method:5s} N={N:6d} {label:7s} {best:8.4f} s  status {r.status} steps {r.n_accepted} nfev {r.nfev} njev {r.njev} nlu {r.nlu}"
                  f" events {[len(e) for e in r.t_events] if r.t_events is not None else None}", flush=True)
            eq.close()
        d = np.max(np.abs(res["pcr"].y_final - res["cr+pcr"].y_final))
        line = f"{method:5s} N={N:6d} max |cr+pcr - pcr| {d:.2e}"
        if N <= ORACLE_MAX_N:
            f = orc.radau if method == "radau" else orc.bdf
            t0 = time.time()
            y, st, *_ = f(orc.params_from_dict(p), N, y0, 0.0, 1.0, 1e-6, 1e-3, 1e-3)
            line += (f" | oracle {time.time() - t0:7.3f} s steps {st.n_accepted} nfev {st.nfev} njev {st.njev} nlu {st.nlu}"
                     f" | max |pcr - oracle| {np.max(np.abs(res['pcr'].y_final - y)):.2e}  max |cr+pcr - oracle| {np.max(np.abs(res['cr+pcr'].y_final - y)):.2e}")
        print(line, flush=True)
