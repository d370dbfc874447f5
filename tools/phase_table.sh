#!/bin/bash
# VALU instructions per RHS evaluation by phase: ablation builds of tools/rk4_lab.hip (every evaluation on its full path), SQ_INSTS_VALU per wave
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd /tmp && export TMPDIR=/tmp && cd $ROOT
out=$ROOT/gpurun_out/r04_s9/phase; mkdir -p $out
for N in 1048576 16384; do for ab in NONE LOG RCP EXP POW POWA FV CORE; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS -d $out/p_${ab}_$N --output-format csv -- tools/lab_bin/rk4_abl_$ab $N 40 1 > $out/p_${ab}_$N.log 2>&1
  python3 - $out/p_${ab}_$N $ab $N <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(float)
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "rk4_fused_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]]+=float(r["Counter_Value"])
w=acc["SQ_WAVES"] or 1
print("N=%s ablate %-5s per wave per evaluation (16 per launch): VALU %.1f  SALU %.1f  LDS %.1f" % (sys.argv[3], sys.argv[2], acc["SQ_INSTS_VALU"]/w/16, acc["SQ_INSTS_SALU"]/w/16, acc["SQ_INSTS_LDS"]/w/16))
PY
  rm -rf $out/p_${ab}_$N
done; done
