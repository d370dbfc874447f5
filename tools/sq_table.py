#!/usr/bin/env python3
"""Per-kernel averages of the SQ counter passes written by tools/profile_sq.sh / profile_round.sh:  sq_table.py <outdir> <tag>"""
import csv
import glob
import os
import sys
from collections import defaultdict

out, tag = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for sub in glob.glob(os.path.join(out, f"pmc_{tag}_SQ*")):
    for f in glob.glob(os.path.join(sub, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            a = acc[row["Kernel_Name"]][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
for k, cs in acc.items():
    if "marl::" not in k or max(v[1] for v in cs.values()) < 2:
        continue
    m = {c: v[0] / v[1] for c, v in cs.items()}
    print(k.replace("void ", "")[:100], "dispatches", max(v[1] for v in cs.values()))
    w = m.get("SQ_WAVES", 0)
    line = {c: round(v, 1) for c, v in m.items()}
    if w:
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
            if c in m:
                line[c + "/wave"] = round(m[c] / w, 1)
        if "SQ_WAVE_CYCLES" in m:
            line["wave_cycles/wave(x4)"] = round(4 * m["SQ_WAVE_CYCLES"] / w, 1)
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS"):
            if c in m and "SQ_WAVE_CYCLES" in m:
                line[c + "/WAVE_CYCLES"] = round(m[c] / m["SQ_WAVE_CYCLES"], 3)
    if "SQ_ACTIVE_INST_VALU" in m and "SQ_INSTS_VALU" in m:
        line["cycles_per_valu"] = round(4 * m["SQ_ACTIVE_INST_VALU"] / m["SQ_INSTS_VALU"], 2)
    print("   ", line)
