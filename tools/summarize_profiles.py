#!/usr/bin/env python3
"""Condense one tools/profile_round.sh output directory into the summaries committed under profiles/.

    python tools/summarize_profiles.py gpurun_out/r01z profiles r01

FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts half of the bytes of coalesced streaming
loads (MI355X_MICROARCH.md, HBM section) - the factor is calibrated in the same run on convert_kernel<0,1>, which
reads exactly 5*8*N bytes with the same 8-byte-per-lane loads."""
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
    """kernel -> counter -> (sum, dispatches)"""
    files = glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0, 0.0]))   # sum, dispatches, max
    for f in files:
        for row in csv.DictReader(open(f)):
            a = acc[row["Kernel_Name"]][row["Counter_Name"]]
            v = float(row["Counter_Value"])
            a[0] += v
            a[1] += 1
            a[2] = max(a[2], v)
    return acc


out = {"command": "python bench.py --no-cpu-baseline --steps 400  (rocprofv3 --pmc ..., one pass per line of tools/profile_round.sh)",
       "units": __doc__.split("\n\n")[2].replace("\n", " "), "kernels": {}}
tables = {c: counter_table("pmc_" + c) for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ1", "SQ2")}
cal = None
for k, v in tables["FETCH_SIZE"].items():
    if "convert_kernel<0, 1>" in k:
        cal = (5 * 8 * N / 1024.0) / v["FETCH_SIZE"][2]   # the N = 2^20 dispatch (the default bench also converts a small grid)
out["fetch_calibration_factor"] = cal
for k in tables["FETCH_SIZE"]:
    if "marl::" not in k or not ("rk4_fused" in k or "convert_kernel<0, 1>" in k):
        continue
    f = tables["FETCH_SIZE"][k]["FETCH_SIZE"]
    w = tables["WRITE_SIZE"][k]["WRITE_SIZE"]
    conv = "convert_kernel" in k
    e = {"FETCH_SIZE_KiB": f[2] if conv else f[0] / f[1], "WRITE_SIZE_KiB": w[2] if conv else w[0] / w[1], "dispatches": f[1]}
    e["hbm_read_bytes_per_launch"] = e["FETCH_SIZE_KiB"] * 1024 * (cal or 2.0)
    e["hbm_write_bytes_per_launch"] = e["WRITE_SIZE_KiB"] * 1024
    e["hbm_bytes_per_launch"] = e["hbm_read_bytes_per_launch"] + e["hbm_write_bytes_per_launch"]
    if "rk4_fused_kernel<256, 1, 1, 4" in k:
        e["algorithmic_bytes_per_launch"] = 80 * N * 4
        sq = {}
        for t in ("SQ1", "SQ2"):
            for c, (s, n, _) in tables[t].get(k, {}).items():
                sq[c] = s / n
        if sq:
            waves = sq.get("SQ_WAVES", 0)
            if waves and "SQ_INSTS_VALU" in sq:
                sq["valu_per_wave_per_step"] = sq["SQ_INSTS_VALU"] / waves / 4
                sq["valu_per_rhs_evaluation"] = sq["valu_per_wave_per_step"] / 4
            if "SQ_ACTIVE_INST_VALU" in sq and "SQ_INSTS_VALU" in sq:
                sq["cycles_per_valu"] = 4 * sq["SQ_ACTIVE_INST_VALU"] / sq["SQ_INSTS_VALU"]
                sq["valu_busy_cycles_per_simd_per_launch"] = 4 * sq["SQ_ACTIVE_INST_VALU"] / 1024
            e["sq_counters"] = sq
    out["kernels"][k.replace("void ", "")] = e
json.dump(out, open(os.path.join(dst, f"{tag}_pmc_hbm_traffic.json"), "w"), indent=1)

stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(dst, f"{tag}_rocprofv3_kernel_stats_bench_default.csv"))
for f in glob.glob(os.path.join(src, "bench_*.json")):
    line = open(f).read().strip().splitlines()
    if line:
        open(os.path.join(dst, f"{tag}_{os.path.basename(f)}"), "w").write(line[-1] + "\n")
        j = json.loads(line[-1])
        print(os.path.basename(f), "%.4e" % j["value"], j.get("roofline", {}).get("frac"))
for k, e in out["kernels"].items():
    print(k, {a: (round(b, 1) if isinstance(b, float) else b) for a, b in e.items() if a != "sq_counters"})
    if "sq_counters" in e:
        print("   ", {a: round(b, 2) for a, b in e["sq_counters"].items()})
