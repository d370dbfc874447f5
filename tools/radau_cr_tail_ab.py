"""Cyclic-reduction tail launches on / off, Radau Scenario A over grid sizes (needs a GPU):  python3 tools/radau_cr_tail_ab.py  (any working directory)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
from common import scenario
from marlpde_amd.LHeureux_model import LMAHeureuxPorosityDiff
for N in (2048, 4000, 16000, 64000, 100001):
    p = scenario("A", N)
    y0 = np.concatenate([np.full(N, p[k]) for k in ("CAIni", "CCIni", "cCaIni", "cCO3Ini", "PhiIni")])
    eq = LMAHeureuxPorosityDiff.from_scenario(p, device=0)
    ref = None
    for tail in (0, 1, 0, 1):
        eq.set_option("radau_cr_tail", tail)
        eq.integrate_radau(y0, (0.0, 1e-4), 1e-6, 1e-3, 1e-3, events=False)
        ts = []
        for rep in range(5):
            t0 = time.perf_counter(); r = eq.integrate_radau(y0, (0.0, 1.0), 1e-6, 1e-3, 1e-3); ts.append(time.perf_counter() - t0)
        same = "" if ref is None else ("bit-identical" if np.array_equal(ref, r.y_final) else "DIFFERENT max %.3e" % np.max(np.abs(ref - r.y_final)))
        ref = r.y_final if ref is None else ref
        print("radau N", N, "tail", tail, "min %.5f median %.5f s" % (min(ts), sorted(ts)[2]), (r.nfev, r.njev, r.nlu, r.n_accepted), r.status, same, flush=True)
    eq.close()
