"""Shared helpers for the test-suite (scenario table of the golden vectors, error measures)."""
import os
from dataclasses import asdict

import numpy as np

from marlpde_amd.parameters import Map_Scenario

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# scenario overrides used by oracle/make_goldens.py (the three the reference's tests use + a stiff one)
SCENARIOS = {
    "default": {},
    "A": {"Phi0": 0.6, "PhiIni": 0.5, "PhiNR": 0.6},              # tests/Regression_test/test_regression.py:43-44
    "matlab": {"Phi0": 0.5, "PhiIni": 0.5, "PhiNR": 0.5, "k3": 0.01, "k4": 0.01},  # :114-116
    "stiffphi": {"b": 0.0005 * 50},
}


def scenario(name, N=200, fv=1, **extra):
    return asdict(Map_Scenario()) | SCENARIOS[name] | {"N": N, "FV_switch": fv} | extra


def parse_key(key):
    s, st, fv, N = str(key).split("|")
    return s, st, int(fv[2:]), int(N[1:])


def rel_to_max(a, b):
    """max |a-b| per field, relative to that field's max |b| (hard part 6)."""
    a, b = np.asarray(a).reshape(5, -1), np.asarray(b).reshape(5, -1)
    scale = np.max(np.abs(b), axis=1, keepdims=True)
    scale = np.where(scale > 0, scale, 1.0)
    return float(np.max(np.abs(a - b) / scale))


def synthetic_state(p, N, amplitude=0.01, waves=8):
    """The bench's deterministic smooth state: initial values x (1 + a sin(2 pi k x / L))."""
    L = p["max_depth"] / p["Xstar"]
    x = (np.arange(N) + 0.5) * (L / N)
    y = np.stack([np.full(N, p[k]) for k in ("CAIni", "CCIni", "cCaIni", "cCO3Ini", "PhiIni")])
    return (y * (1.0 + amplitude * np.sin(2 * np.pi * waves * x / L))).ravel()


def noisy_state(p, N, seed=0, sigma=0.05):
    y = np.stack([np.full(N, p[k]) for k in ("CAIni", "CCIni", "cCaIni", "cCO3Ini", "PhiIni")])
    return (y * (1.0 + sigma * np.random.default_rng(seed).standard_normal((5, N)))).ravel()


def record_parity(kind, key, got, want, same):
    """Append one line to the implicit-path parity report: which (case, grouping, solver) combinations took
    EVERY decision scipy took.  Written to gpurun_out/implicit_parity_report.jsonl when that directory is writable (the GPU box
    merges it back; a copy of a full run is committed as profiles/r03_implicit_parity_report.jsonl) and printed (pytest -rA)."""
    import json
    line = json.dumps({"solver": kind, "case": key, "got_nfev_njev_nlu_steps": list(map(int, got)), "scipy_nfev_njev_nlu_steps": list(map(int, want)),
                       "every_decision_as_scipy": bool(same)})
    print("PARITY", line)
    out = os.path.join(os.path.dirname(GOLDEN), "..", "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "implicit_parity_report.jsonl"), "a") as f:
            f.write(line + "\n")
    except OSError:
        pass
