#!/bin/bash
# Per-wave instruction counts of one kernel for several library builds, on one box:
#   tools/pmc_ab.sh <outdir> "<bench args>" <kernel substring> lib1.so lib2.so ...
out=$1; args=$2; kern=$3; shift 3
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for lib in "$@"; do
  export MARL_HIP_LIBRARY=$PWD/$lib
  tag=$(basename $lib .so)
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS -d $out/pmc_$tag --output-format csv -- python3 bench.py --no-cpu-baseline --no-extras $args > $out/pmc_$tag.log 2>&1
  python3 - $out/pmc_$tag $tag "$kern" <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(float)
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[3] in r["Kernel_Name"]: acc[r["Counter_Name"]]+=float(r["Counter_Value"])
w=acc["SQ_WAVES"] or 1
print(sys.argv[2], "per wave: " + "  ".join("%s %.1f" % (k[3:], v/w) for k,v in sorted(acc.items()) if k!="SQ_WAVES"), " waves %d" % w)
PY
  rm -rf $out/pmc_$tag
done
