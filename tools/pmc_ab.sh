#!/bin/bash
# Per-wave instruction counts of one kernel for several library builds, on one box:
#   tools/pmc_ab.sh <outdir> "<bench args>" <kernel substring> lib1.so lib2.so ...
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
[ $# -ge 4 ] || { echo "usage: tools/pmc_ab.sh <outdir> \"<bench args>\" <kernel substring> lib1.so [lib2.so ...]" >&2; exit 2; }
out=$1; args=$2; kern=$3; shift 3
mkdir -p "$ROOT/$out"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
for lib in "$@"; do
  [ -f "$ROOT/$lib" ] || { echo "pmc_ab.sh: no such library: $lib" >&2; exit 2; }
  export MARL_HIP_LIBRARY=$ROOT/$lib
  tag=$(basename $lib .so)
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS -d $out/pmc_$tag --output-format csv -- python3 bench.py --no-cpu-baseline --no-extras $args > $out/pmc_$tag.log 2>&1
  python3 - $out/pmc_$tag $tag "$kern" <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(float)
files=glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True)
if not files: sys.exit("pmc_ab.sh: no counter CSV under "+sys.argv[1]+" (see the .log beside it)")
for f in files:
    for r in csv.DictReader(open(f)):
        if sys.argv[3] in r["Kernel_Name"]: acc[r["Counter_Name"]]+=float(r["Counter_Value"])
w=acc["SQ_WAVES"] or 1
print(sys.argv[2], "per wave: " + "  ".join("%s %.1f" % (k[3:], v/w) for k,v in sorted(acc.items()) if k!="SQ_WAVES"), " waves %d" % w)
PY
  rm -rf $out/pmc_$tag
done
