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
os.makedirs(dst, exist_ok=True)
N = 1 << 20


# A dispatch that found nothing to do: rk45_run / marl_slab_run enqueue `poll_interval` attempts per batch, and the attempt kernel
# and the kernels that follow it (record reduction, pack / unpack, control) have nothing to do once the controller has left
# ST_RUNNING (budget reached, pause for a t_eval sample or an event).  Such dispatches must not enter a per-launch average
# (100 of 320 counter dispatches and 92 of 512 traced calls of the rk45_single profile were no-ops; the
# averages were 17 - 30 % too low).  Criterion: an rk45_attempt_kernel dispatch whose work-proportional quantity (instructions,
# wave-cycles, bytes: < 5 % of the kernel's median; duration: < 25 %) is a no-op, and so is every follower kernel dispatched
# between it and the next attempt.  Nothing else is ever dropped.
NOOP_FRACTION, NOOP_FRACTION_DURATION = 0.05, 0.25
PRIMARY = ("SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "FETCH_SIZE", "WRITE_SIZE")
LEADER = "rk45_attempt_kernel"
FOLLOWERS = ("reduce_chunks_kernel", "rk45_control_kernel", "slab_reduce_pack_kernel", "slab_unpack_control_kernel")


def median(v):
    v = sorted(v)
    return v[len(v) // 2] if v else 0.0


def mark_noops(seq, fraction):
    """seq: [(dispatch id, kernel name, work)] of one process -> set of dispatch ids that did nothing"""
    floor = {}
    for k in {k for _, k, _ in seq if LEADER in k}:
        floor[k] = fraction * median([w for _, kk, w in seq if kk == k])
    noop, idle = set(), False
    for d, k, w in sorted(seq):
        if LEADER in k:
            idle = w < floor[k]
            if idle:
                noop.add(d)
        elif any(f in k for f in FOLLOWERS):
            if idle:
                noop.add(d)
        # any other kernel: leaves `idle` as it is (dense-output replays between a pause and the resume do not restart the loop)
    return noop


def counter_table(sub):
    """kernel -> counter -> [sum, dispatches, max], over the dispatches that did work; + kernel -> "_noop" -> dropped dispatches"""
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0, 0.0]))
    for f in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
        disp = defaultdict(dict)    # dispatch id -> counter -> value
        name = {}
        for row in csv.DictReader(open(f)):
            d = int(row["Dispatch_Id"])
            name[d] = row["Kernel_Name"]
            disp[d][row["Counter_Name"]] = disp[d].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])   # (one row per counter instance)
        prim = next((c for c in PRIMARY if all(c in v for d, v in disp.items() if LEADER in name[d])), PRIMARY[0])
        noop = mark_noops([(d, name[d], v.get(prim, 0.0)) for d, v in disp.items()], NOOP_FRACTION)
        for d, v in disp.items():
            if d in noop:
                acc[name[d]]["_noop"][1] += 1
                continue
            for c, x in v.items():
                a = acc[name[d]][c]
                a[0] += x
                a[1] += 1
                a[2] = max(a[2], x)
    return acc


def trace_stats(trace_csv, out_csv):
    """rocprofv3's kernel_stats.csv recomputed from its kernel_trace.csv WITHOUT the no-op dispatches (same columns + how many
    were dropped): the committed average of a kernel is the mean of its real launches."""
    seq = [(int(r["Dispatch_Id"]), r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(trace_csv))]
    noop = mark_noops(seq, NOOP_FRACTION_DURATION)
    dur, dropped = defaultdict(list), defaultdict(int)
    for d, k, w in seq:
        if d in noop:
            dropped[k] += 1
        else:
            dur[k].append(w)
    table, total = [], 0
    for k, real in dur.items():
        n, s = len(real), sum(real)
        mean = s / n
        sd = (sum((x - mean) ** 2 for x in real) / n) ** 0.5
        table.append([k, n, s, mean, min(real), max(real), sd, dropped[k]])
        total += s
    table.sort(key=lambda r: -r[2])
    with open(out_csv, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev", "NoopCallsDropped"])
        for k, n, s, mean, lo, hi, sd, dr in table:
            w.writerow([k, n, s, round(mean, 1), round(100.0 * s / total, 2), lo, hi, round(sd, 1), dr])


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
        e = {"workload": wl, "FETCH_SIZE_KiB": f[0] / f[1], "WRITE_SIZE_KiB": w[0] / w[1], "dispatches": fe[k]["FETCH_SIZE"][1],
             "noop_dispatches_dropped": fe[k]["_noop"][1] if "_noop" in fe[k] else 0, "fetch_calibration_factor": cal}
        e["hbm_read_bytes_per_launch"] = e["FETCH_SIZE_KiB"] * 1024 * (cal or 2.0)
        e["hbm_write_bytes_per_launch"] = e["WRITE_SIZE_KiB"] * 1024
        e["hbm_bytes_per_launch"] = e["hbm_read_bytes_per_launch"] + e["hbm_write_bytes_per_launch"]
        e["algorithmic_bytes_per_launch"] = alg
        e["steps_per_launch"] = spl
        hbm["kernels"][k.replace("void ", "")] = e
json.dump(hbm, open(os.path.join(dst, f"{tag}_pmc_hbm_traffic.json"), "w"), indent=1)

# ---- SQ counters per workload and kernel ---------------------------------------------------------------------------
sq = {"command": "rocprofv3 --pmc <SQ set> -- python3 bench.py --no-cpu-baseline --no-extras ... (tools/profile_round.sh); SQ_WAVE_CYCLES, "
                 "SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles; averages per dispatch THAT DID WORK (dispatches of the attempt / "
                 "reduce / control kernels that returned at once because the controller had stopped are dropped: noop_dispatches_dropped)",
      "workloads": {}}
# grid-point-steps (attempted steps for RK45) that `n` counted dispatches of the dominant kernel of a workload cover - follows the
# bench arguments of tools/profile_round.sh (default: settle call of 2000 steps, then calls of 400; n65536: 16 steps per launch;
# sweeps: one warm-up call of 5 and one timed call of 200 attempts over 4096 x 1024 cells)
UNITS = {("default", "rk4_stream_kernel<256, 1, 4"): lambda n: N * (2000 + 400 * (n - 1)),
         ("default_no_reuse", "rk4_stream_kernel<256, 1, 4"): lambda n: N * (2000 + 400 * (n - 1)),
         ("n65536", "rk4_fused_kernel<256, 1, 1, 16"): lambda n: 65536 * 16 * n,
         ("n65536", "rk4_stream_kernel"): lambda n: 65536 * (2000 + 800 * (n - 1)),
         ("rk45_single", "rk45_attempt_kernel"): lambda n: N * n,
         ("sweep_rk45", "rk45_sweep_kernel"): lambda n: 4096 * 1024 * 205 if n == 2 else None,
         ("sweep_rk4", "rk4_sweep_kernel"): lambda n: 4096 * 1024 * 205 if n == 2 else None}
for wl in ("default", "default_no_reuse", "n65536", "rk45_single", "sweep_rk45", "sweep_rk4"):
    acc = defaultdict(dict)
    for sub in glob.glob(os.path.join(src, f"pmc_{wl}_SQ*")):
        for k, cs in counter_table(os.path.basename(sub)).items():
            for c, (s, n, _) in cs.items():
                if c == "_noop":
                    acc[k]["noop_dispatches_dropped"] = max(acc[k].get("noop_dispatches_dropped", 0), n)
                    continue
                acc[k][c] = s / n
                acc[k]["dispatches"] = n
    out = {}
    for k, m in acc.items():
        if "marl" not in k or m.get("dispatches", 0) < 2 or not any(x in k for x in ("rk4_fused", "rk4_stream", "rk45_attempt", "sweep_kernel", "control", "reduce_chunks")):
            continue
        w = m.get("SQ_WAVES", 0)
        if w and "SQ_INSTS_VALU" in m:
            m["valu_per_wave"] = m["SQ_INSTS_VALU"] / w
            m["lds_per_wave"] = m.get("SQ_INSTS_LDS", 0) / w
        units = next((f(m["dispatches"]) for (uw, uk), f in UNITS.items() if uw == wl and uk in k), None)
        if units and "SQ_INSTS_VALU" in m:   # wave-instructions per grid-point-step: bench.py's valu_issue_frac multiplies by its own rate
            m["valu_insts_per_grid_point_step"] = m["SQ_INSTS_VALU"] * m["dispatches"] / units
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

# ---- the implicit path: per kernel of the single Radau runs (N = 200 / 16 000 / 64 000), HBM bytes and VALU instructions against time ----
def totals(sub):
    """kernel -> counter -> sum over ALL dispatches (these kernels have no no-op launches)"""
    acc = defaultdict(lambda: defaultdict(float))
    for f in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"]][row["Counter_Name"]] += float(row["Counter_Value"])
            acc[row["Kernel_Name"]]["_ns_" + row["Counter_Name"]] += int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
            acc[row["Kernel_Name"]]["_n_" + row["Counter_Name"]] += 1
    return acc


imp = {"command": "rocprofv3 --kernel-trace --stats | --pmc FETCH_SIZE | WRITE_SIZE | SQ set -- python3 tools/radau_profile.py single   (Scenario A to T* "
                  "with marl_integrate_radau at N = 200, 16 000 and 64 000, one after another; tools/profile_round.sh)",
       "units": "sums over every dispatch of the kernel in the run; durations from the kernel trace of the --stats pass; FETCH_SIZE doubled "
                "(gfx950: it counts half of the bytes of streaming loads - calibrated on convert_kernel in the explicit passes); "
                "hbm_gbs = bytes / kernel time; valu_issue_frac = VALU wave-instructions / kernel time / (1024 SIMDs x 2.4 GHz / 4)",
       "kernels": {}}
tr = glob.glob(os.path.join(src, "stats_radau_single", "**", "*kernel_trace.csv"), recursive=True)
if tr:
    dur, calls = defaultdict(int), defaultdict(int)
    for r in csv.DictReader(open(tr[0])):
        dur[r["Kernel_Name"]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        calls[r["Kernel_Name"]] += 1
    fe, wr, sq1 = totals("pmc_radau_single_FETCH_SIZE"), totals("pmc_radau_single_WRITE_SIZE"), totals("pmc_radau_single_SQ1")
    total_ns = sum(dur.values())
    for k in sorted(dur, key=lambda x: -dur[x]):
        if "marl::" not in k:
            continue
        e = {"calls": calls[k], "total_us": dur[k] / 1e3, "avg_us": dur[k] / 1e3 / calls[k], "share_of_gpu_time": dur[k] / total_ns}
        if k in fe and k in wr:
            rd, wrb = fe[k]["FETCH_SIZE"] * 1024 * 2.0, wr[k]["WRITE_SIZE"] * 1024
            e.update(hbm_read_bytes=rd, hbm_write_bytes=wrb, hbm_gbs=(rd + wrb) / dur[k], hbm_frac_of_8TBs=(rd + wrb) / dur[k] / 8000.0)
        if k in sq1 and "SQ_INSTS_VALU" in sq1[k]:
            e.update(valu_wave_insts=sq1[k]["SQ_INSTS_VALU"], valu_issue_frac=sq1[k]["SQ_INSTS_VALU"] / (dur[k] * 1e-9) / (256 * 4 * 2.4e9 / 4))
        imp["kernels"][k.replace("void ", "")[:110]] = {a: (round(b, 4) if isinstance(b, float) else b) for a, b in e.items()}
    json.dump(imp, open(os.path.join(dst, f"{tag}_implicit_kernels.json"), "w"), indent=1)
for name in ("stats_radau_single.log", "stats_radau_bdf_sweep.log"):
    f = os.path.join(src, name)
    if os.path.exists(f):
        keep = [ln for ln in open(f).read().splitlines() if ln.startswith(("radau", "bdf"))]
        open(os.path.join(dst, f"{tag}_{name.replace('stats_', 'timing_')}"), "w").write("\n".join(keep) + "\n")

# ---- kernel-trace statistics and bench lines -------------------------------------------------------------------------
for d in glob.glob(os.path.join(src, "stats_*")):
    if not os.path.isdir(d):
        continue
    wl = os.path.basename(d)[len("stats_"):]
    st = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    tr = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if tr:     # recomputed from the trace without no-op dispatches; rocprofv3's own table beside it
        trace_stats(tr[0], os.path.join(dst, f"{tag}_rocprofv3_kernel_stats_{wl}.csv"))
        if st:
            shutil.copy(st[0], os.path.join(dst, f"{tag}_rocprofv3_kernel_stats_{wl}_raw.csv"))
    elif st:
        shutil.copy(st[0], os.path.join(dst, f"{tag}_rocprofv3_kernel_stats_{wl}.csv"))
for f in sorted(glob.glob(os.path.join(src, "bench_*.json"))):
    line = open(f).read().strip().splitlines()
    if line:
        open(os.path.join(dst, f"{tag}_{os.path.basename(f)}"), "w").write(line[-1] + "\n")
        j = json.loads(line[-1])
        print(os.path.basename(f), "%.4e" % j["value"], "frac %.3f" % j.get("roofline", {}).get("frac", 0))
# ---- traced against untraced: the bench line of the run that was traced beside the same command without the profiler ----------------
side = {}
for f in sorted(glob.glob(os.path.join(src, "traced_*.json"))):
    wl = os.path.basename(f)[len("traced_"):-len(".json")]
    line = [ln for ln in open(f).read().strip().splitlines() if ln.startswith("{")]
    if not line:
        continue
    t = json.loads(line[-1])
    e = {"traced_ms_per_step": t["ms_per_step"], "traced_value": t["value"], "steps": t["steps"]}
    for cand in (f"bench400_{wl}.json", f"bench_{wl}.json"):   # the same command untraced (default / n65536: the bench line itself)
        g = os.path.join(src, cand)
        if os.path.exists(g) and open(g).read().strip():
            u = json.loads(open(g).read().strip().splitlines()[-1])
            e.update(untraced_ms_per_step=u["ms_per_step"], untraced_value=u["value"], untraced_steps=u["steps"], untraced_file=cand)
            break
    side[wl] = e
if side:
    json.dump({"what": "ms_per_step of the traced run (whose per-kernel times the *_rocprofv3_kernel_stats_<workload>.csv tables hold) beside the same "
                       "command untraced, so that profiles/ alone reproduces each bench line (tools/profile_round.sh)", "workloads": side},
              open(os.path.join(dst, f"{tag}_traced_vs_untraced.json"), "w"), indent=1)
for k, e in hbm["kernels"].items():
    print(k[:70], {a: (round(b, 1) if isinstance(b, float) else b) for a, b in e.items()})
