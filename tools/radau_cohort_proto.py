#!/usr/bin/env python3
"""Prototype: a Radau sweep split into K cohorts (contiguous instance ranges), each its own context + stream + host thread."""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
from common import scenario
from marlpde_amd.LHeureux_model import LMAHeureuxPorosityDiff
N = 200
base = scenario("default", N)
def grid(B):
    k = max(1, round(B ** (1 / 3)))
    inst = []
    for i in range(B):
        d = {"Phi0": 0.5 + 0.2 * ((i % k) / max(k - 1, 1)), "PhiIni": 0.5 + 0.2 * (((i // k) % k) / max(k - 1, 1)),
             "k3": 10 ** (-2 + ((i // (k * k)) % k) / max(k - 1, 1))}
        d["PhiNR"] = d["PhiIni"]; d["k4"] = d["k3"]
        inst.append(d)
    return inst
for B in [int(a) for a in sys.argv[1:]] or [512, 4096]:
    inst = grid(B)
    y0 = np.stack([np.concatenate([np.full(N, (base | d)[q]) for q in ("CAIni", "CCIni", "cCaIni", "cCO3Ini", "PhiIni")]) for d in inst])
    ref = None
    for K in (1, 4, 8, 16, 32):
        bounds = [(B * k) // K for k in range(K + 1)]
        eqs, streams, yds = [], [], []
        for k in range(K):
            eq = LMAHeureuxPorosityDiff.from_scenario(base, device=0, instances=inst[bounds[k]:bounds[k + 1]])
            s = torch.cuda.Stream()
            eq.use_stream(s.cuda_stream)
            eqs.append(eq); streams.append(s)
        for rep in range(2):
            yds = [torch.from_numpy(y0[bounds[k]:bounds[k + 1]]).cuda() for k in range(K)]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            def work(k):
                return eqs[k].sweep_radau_device(yds[k].data_ptr(), (0.0, 1.0), 1e-6, 1e-3, 1e-3)
            with ThreadPoolExecutor(max_workers=K) as pool:
                res = [r for part in pool.map(work, range(K)) for r in part]
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        stats = [(r.nfev, r.njev, r.nlu) for r in res]
        if ref is None:
            ref = stats
        print(f"B={B} K={K:2d}: {dt:.3f} s  same statistics as K=1: {stats == ref}  nfev max {max(s[0] for s in stats)}", flush=True)
        for eq in eqs:
            eq.close()
