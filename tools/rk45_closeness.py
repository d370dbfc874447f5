#!/usr/bin/env python3
"""How close do the accept / reject decisions of the explicit path sit to their knife edge?

The fused Dormand-Prince kernels are pinned by tolerance plus EQUAL decision sequences on three scipy-on-the-reference trajectories
(tests/golden/rk45_traj_*.npz; tests/test_gpu_parity.py::test_rk45_reproduces_scipy_trajectory), not bit by bit: their evaluation
order differs from the stand-alone RHS in the last bits (csrc/marl_math.h, LEGACY = false).  A decision flips when an attempt's
error norm crosses 1, so the margin that every further instruction cut has to respect is min |err_norm - 1| over all attempts.
This script re-walks each golden trajectory on the CPU oracle's RHS with scipy's step logic (rk.py:111-176 restated in numpy),
checks that it takes scipy's own steps, and writes the distances:

    python3 tools/rk45_closeness.py [out.json]        (CPU only; default out: profiles/r04_rk45_closeness.json)

The GPU's error norms differ from these by ~1e-13 relative (RHS agreement 2e-15 of the field maximum, amplified by 1 / rtol),
so a minimum distance of 1e-4 is a margin of nine orders of magnitude.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

C = np.array([0, 1 / 5, 3 / 10, 4 / 5, 8 / 9, 1])
A = [[], [1 / 5], [3 / 40, 9 / 40], [44 / 45, -56 / 15, 32 / 9], [19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729],
     [9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656]]
B = np.array([35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84])
E = np.array([-71 / 57600, 0, 71 / 16695, -71 / 1920, 17253 / 339200, -22 / 525, 1 / 40])


def walk(rhs, y0, t0, t1, first_step, rtol, atol):
    """scipy's RK45 loop; returns (accepted step times, [(err_norm, accepted)] of every attempt)."""
    t, y, f, h_abs = t0, y0.copy(), rhs(y0), first_step
    times, attempts, n = [t0], [], y0.size
    while t < t1:
        min_step = 10 * abs(np.nextafter(t, np.inf) - t)
        h_abs = max(h_abs, min_step)
        rejected = False
        while True:
            if h_abs < min_step:
                raise RuntimeError("step size too small")
            t_new = min(t + h_abs, t1)
            h = t_new - t
            h_abs = abs(h)
            K = np.empty((7, n))
            K[0] = f
            for s in range(1, 6):
                K[s] = rhs(y + h * (K[:s].T @ np.array(A[s])))
            y_new = y + h * (K[:6].T @ B)
            K[6] = rhs(y_new)
            scale = atol + np.maximum(np.abs(y), np.abs(y_new)) * rtol
            err = np.linalg.norm((K.T @ E) * h / scale) / n ** 0.5
            attempts.append((float(err), bool(err < 1)))
            if err < 1:
                factor = 10.0 if err == 0 else min(10.0, 0.9 * err ** -0.2)
                if rejected:
                    factor = min(1.0, factor)
                h_abs *= factor
                break
            h_abs *= max(0.2, 0.9 * err ** -0.2)
            rejected = True
        t, y, f = t_new, y_new, K[6]
        times.append(t)
    return np.array(times), attempts


def report(traj):
    from common import GOLDEN, scenario
    from oracle import oracle as orc
    g = np.load(f"{GOLDEN}/{traj}.npz")
    name, N = traj.split("_")[2], int(traj.split("N")[-1])
    P = orc.params_from_dict(scenario(name, N))
    times, att = walk(lambda y: orc.rhs(P, N, y), g["y0"], float(g["t_span"][0]), float(g["t_span"][1]), float(g["first_step"]), float(g["rtol"]),
                      float(g["atol"]))
    same = len(times) == len(g["step_times"]) and bool(np.allclose(times, g["step_times"], rtol=1e-9, atol=0))
    d = np.array([abs(e - 1.0) for e, _ in att])
    return {"trajectory": traj, "N": N, "attempts": len(att), "accepted": int(sum(a for _, a in att)), "rejected": int(sum(not a for _, a in att)),
            "walks_scipys_steps": same, "min_abs_err_norm_minus_1": float(d.min()), "attempt_of_minimum": int(d.argmin()),
            "attempts_within_1e-3": int((d < 1e-3).sum()), "attempts_within_1e-6": int((d < 1e-6).sum()),
            "largest_accepted_err_norm": float(max(e for e, a in att if a)), "smallest_rejected_err_norm": float(min([e for e, a in att if not a] or [np.inf]))}


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r04_rk45_closeness.json")
    rows = [report(t) for t in ("rk45_traj_A_N200", "rk45_traj_default_N200", "rk45_traj_A_N64")]
    json.dump({"what": __doc__.split("\n\n")[0] + " (tools/rk45_closeness.py)", "trajectories": rows}, open(out, "w"), indent=1)
    for r in rows:
        print(r)
