"""Per-kernel totals and inter-kernel gaps from a rocprofv3 --kernel-trace run.

    python3 tools/trace_levels.py <rocprofv3 output directory>     (the directory given to rocprofv3 -d; any working directory)
"""
import collections
import csv
import glob
import sys

if len(sys.argv) != 2:
    sys.exit(__doc__)
found = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
if not found:
    sys.exit(f"trace_levels.py: no *kernel_trace.csv under {sys.argv[1]}")
f = found[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
agg = collections.defaultdict(lambda: [0, 0.0])
gaps = collections.defaultdict(lambda: [0, 0.0])
prev_end = None
for r in rows:
    name = r["Kernel_Name"].split("(")[0].split("::")[-1]
    key = (name, int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"]))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg[key][0] += 1; agg[key][1] += d
    if prev_end is not None:
        g = (int(r["Start_Timestamp"]) - prev_end) / 1e3
        if g < 50: gaps[name][0] += 1; gaps[name][1] += g
    prev_end = int(r["End_Timestamp"])
tot = sum(v[1] for v in agg.values())
print("total kernel time %.1f ms, span %.1f ms" % (tot / 1e3, (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e6))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print("%-28s grid %9d calls %6d avg %8.2f us total %8.2f ms" % (k[0], k[1], v[0], v[1] / v[0], v[1] / 1e3))
print("gap before kernel (us, gaps < 50 us):")
for k, v in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:12]:
    print("%-28s n %6d avg %6.2f total %8.2f ms" % (k, v[0], v[1] / v[0], v[1] / 1e3))
