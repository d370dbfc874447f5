#!/usr/bin/env python3
"""Condense one tools/profile_round.sh output directory into the summaries committed under profiles/.

    python tools/summarize_profiles.py gpurun_out/r02p profiles r02

Writes  <tag>_bench_<workload>.json  (the bench JSON line of every workload),  <tag>_rocprofv3_kernel_stats_<workload>.csv  (the
rocprofv3 --kernel-trace --stats table of every workload),  <tag>_pmc_hbm_traffic.json  (HBM bytes per launch of the dominant
kernels) and  <tag>_pmc_sq_counters.json  (SQ instruction / wait counters per kernel and workload, with derived figures).

FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts half of the bytes of coalesced streaming loads
(MI355X_MICROARCH.md, HBM section) - the factor is calibrated in the same run on convert_kernel<0,1>, which reads exactly
5*8*N bytes with the same 8-byte-per-lane loads."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
N = 1 << 20


def counter_table(sub):
    """kernel -> counter -> [sum, dispatches, max]"""
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0, 0.0]))
    for f in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            a = acc[row["Kernel_Name"]][row["Counter_Name"]]
            v = float(row["Counter_Value"])
            a[0] += v
            a[1] += 1
            a[2] = max(a[2], v)
    return acc


# ---- HBM traffic of the dominant kernels -----------------------------------------------------------------------
hbm = {"command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (one counter per pass) -- python3 bench.py --no-cpu-baseline --no-extras ... "
                  "(tools/profile_round.sh)", "units": __doc__.split("\n\n")[3].replace("\n", " "), "kernels": {}}
# (the streamed RK4 kernel: one launch per bench call - bench.py's settle call of 2000 steps, then calls of 400 steps:
#  tools/profile_round.sh passes --steps 400 --warmup 400 - so the counters are summed over the dispatches and divided by the steps)
for wl, pick, alg, spl in (("default", "rk4_stream_kernel<256, 1, 4", 80 * N * 400, 400), ("default", "rk4_fused_kernel<256, 1, 1, 4", 80 * N * 4, 4),
                           ("rk45_single", "rk45_attempt_kernel<256, 1, 1", 160 * N, 1)):
    fe, wr = counter_table(f"pmc_{wl}_FETCH_SIZE"), counter_table(f"pmc_{wl}_WRITE_SIZE")
    cal = None
    for k, v in fe.items():
        if "convert_kernel<0, 1>" in k:
            cal = (5 * 8 * N / 1024.0) / v["FETCH_SIZE"][2]   # the N = 2^20 dispatch
    for k in fe:
        if pick not in k:
            continue
        f, w = fe[k]["FETCH_SIZE"], wr[k]["WRITE_SIZE"]
        if "rk4_stream" in pick:   # per launch of `spl` steps = total / total steps x spl
            f = [f[0] * spl / (2000 + spl * (f[1] - 1)), 1, 0]
            w = [w[0] * spl / (2000 + spl * (w[1] - 1)), 1, 0]
        e = {"workload": wl, "FETCH_SIZE_KiB": f[0] / f[1], "WRITE_SIZE_KiB": w[0] / w[1], "dispatches": fe[k]["FETCH_SIZE"][1], "fetch_calibration_factor": cal}
        e["hbm_read_bytes_per_launch"] = e["FETCH_SIZE_KiB"] * 1024 * (cal or 2.0)
        e["hbm_write_bytes_per_launch"] = e["WRITE_SIZE_KiB"] * 1024
        e["hbm_bytes_per_launch"] = e["hbm_read_bytes_per_launch"] + e["hbm_write_bytes_per_launch"]
        e["algorithmic_bytes_per_launch"] = alg
        e["steps_per_launch"] = spl
        hbm["kernels"][k.replace("void ", "")] = e
json.dump(hbm, open(os.path.join(dst, f"{tag}_pmc_hbm_traffic.json"), "w"), indent=1)

# ---- SQ counters per workload and kernel ---------------------------------------------------------------------------
sq = {"command": "rocprofv3 --pmc <SQ set> -- python3 bench.py --no-cpu-baseline --no-extras ... (tools/profile_round.sh); SQ_WAVE_CYCLES, "
                 "SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles; averages per dispatch", "workloads": {}}
for wl in ("default", "n65536", "rk45_single", "sweep_rk45", "sweep_rk4"):
    acc = defaultdict(dict)
    for sub in glob.glob(os.path.join(src, f"pmc_{wl}_SQ*")):
        for k, cs in counter_table(os.path.basename(sub)).items():
            for c, (s, n, _) in cs.items():
                acc[k][c] = s / n
                acc[k]["dispatches"] = n
    out = {}
    for k, m in acc.items():
        if "marl::" not in k or m.get("dispatches", 0) < 2 or not any(x in k for x in ("rk4_fused", "rk4_stream", "rk45_attempt", "sweep_kernel", "control", "reduce_chunks")):
            continue
        w = m.get("SQ_WAVES", 0)
        if w and "SQ_INSTS_VALU" in m:
            m["valu_per_wave"] = m["SQ_INSTS_VALU"] / w
            m["lds_per_wave"] = m.get("SQ_INSTS_LDS", 0) / w
        if "SQ_ACTIVE_INST_VALU" in m and "SQ_INSTS_VALU" in m:
            m["cycles_per_valu"] = 4 * m["SQ_ACTIVE_INST_VALU"] / m["SQ_INSTS_VALU"]
        if "SQ_WAVE_CYCLES" in m:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU"):
                if c in m:
                    m[c + "_frac_of_wave_cycles"] = m[c] / m["SQ_WAVE_CYCLES"]
        out[k.replace("void ", "")] = {a: (round(b, 4) if isinstance(b, float) else b) for a, b in m.items()}
    if out:
        sq["workloads"][wl] = out
json.dump(sq, open(os.path.join(dst, f"{tag}_pmc_sq_counters.json"), "w"), indent=1)

# ---- kernel-trace statistics and bench lines -------------------------------------------------------------------------
for d in glob.glob(os.path.join(src, "stats_*")):
    if not os.path.isdir(d):
        continue
    wl = os.path.basename(d)[len("stats_"):]
    st = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    if st:
        shutil.copy(st[0], os.path.join(dst, f"{tag}_rocprofv3_kernel_stats_{wl}.csv"))
for f in sorted(glob.glob(os.path.join(src, "bench_*.json"))):
    line = open(f).read().strip().splitlines()
    if line:
        open(os.path.join(dst, f"{tag}_{os.path.basename(f)}"), "w").write(line[-1] + "\n")
        j = json.loads(line[-1])
        print(os.path.basename(f), "%.4e" % j["value"], "frac %.3f" % j.get("roofline", {}).get("frac", 0))
for k, e in hbm["kernels"].items():
    print(k[:70], {a: (round(b, 1) if isinstance(b, float) else b) for a, b in e.items()})
