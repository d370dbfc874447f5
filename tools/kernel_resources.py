#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS / occupancy table from the compiler's resource remarks.

    make -C integrating-diagenetic-equations-using-python_amd/csrc resources > /tmp/resources.txt 2>&1
    python tools/kernel_resources.py /tmp/resources.txt [substring ...]
"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
want = sys.argv[2:]
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
names = [b.split("\n")[0].split(" ")[0] for b in blocks]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
print(f"{'kernel':90s} {'VGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'occ':>4s} {'LDS':>7s}")
for b, d in zip(blocks, dem):
    d = d.split("(")[0].replace("void marl::", "")
    if want and not any(w in d for w in want):
        continue
    vals = [re.search(pat, b).group(1) for pat in (r"VGPRs: (\d+)", r"SGPRs: (\d+)", r"ScratchSize \[bytes/lane\]: (\d+)",
                                                   r"Occupancy \[waves/SIMD\]: (\d+)", r"LDS Size \[bytes/block\]: (\d+)")]
    print(f"{d[:90]:90s} {vals[0]:>5s} {vals[1]:>5s} {vals[2]:>8s} {vals[3]:>4s} {vals[4]:>7s}")
