#!/usr/bin/env python3
"""Per-cycle picture of a Radau sweep from a kernel trace:
    rocprofv3 --kernel-trace -d gpurun_out/rtrace --output-format csv -- python3 tools/radau_sweep_timing.py 512
    python3 tools/radau_sweep_trace.py gpurun_out/rtrace          (any working directory)
Prints, for every launch of radau_wg_kernel in the LAST sweep of the trace: its grid (= min(instances visited, CUs)), its duration and the
time until the next launch of it (the host cycle: list lengths, factorisation launches), in deciles of the sweep - where the wall time of
the sweep goes and who is on its critical path."""
import csv
import glob
import os
import sys

d = sys.argv[1]
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"])))
rows.sort()
wg = [i for i, r in enumerate(rows) if "radau_wg_kernel" in r[2]]
if not wg:
    sys.exit("no radau_wg_kernel in the trace")
t_first, t_last = rows[wg[0]][0], rows[-1][1]
print(f"{len(wg)} launches of radau_wg_kernel over {(t_last - t_first) * 1e-6:.1f} ms")
tot_wg = sum(rows[i][1] - rows[i][0] for i in wg)
print(f"sum of radau_wg_kernel durations {tot_wg * 1e-6:.1f} ms; between them {((t_last - t_first) - tot_wg) * 1e-6:.1f} ms")
n = len(wg)
print(" decile  launches   grid(wgs) min/max    wg kernel us mean/max    gap to next us mean   kernels in gap (mean)  busy in gap us")
for q in range(10):
    sel = wg[q * n // 10:(q + 1) * n // 10]
    if not sel:
        continue
    dur = [(rows[i][1] - rows[i][0]) * 1e-3 for i in sel]
    grid = [rows[i][3] // 1024 for i in sel]
    gaps, cnt, busy = [], [], []
    for i in sel:
        k = wg.index(i)
        nxt = wg[k + 1] if k + 1 < n else len(rows)
        end_next = rows[nxt][0] if nxt < len(rows) else rows[-1][1]
        gaps.append((end_next - rows[i][1]) * 1e-3)
        cnt.append(nxt - i - 1)
        busy.append(sum(rows[j][1] - rows[j][0] for j in range(i + 1, nxt)) * 1e-3)
    print(f"  {q:3d}    {len(sel):6d}     {min(grid):4d} /{max(grid):4d}        {sum(dur) / len(dur):8.1f} /{max(dur):8.1f}        {sum(gaps) / len(gaps):8.1f}"
          f"            {sum(cnt) / len(cnt):6.1f}           {sum(busy) / len(busy):8.1f}")
