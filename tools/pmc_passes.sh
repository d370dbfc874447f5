#!/bin/bash
# counter passes for one bench command: tools/pmc_passes.sh <outdir> "<MARL_HIP_OPTIONS>" "<bench args>" <kernel substring>
ROOT=$(cd "$(dirname "$0")/.." && pwd)
out=$ROOT/$1; opts=$2; args=$3; kern=$4
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $ROOT
export MARL_HIP_OPTIONS="$opts"
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH" "SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32" "SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_WAVES SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $out/p$i --output-format csv -- python3 bench.py --no-cpu-baseline --no-extras $args > $out/p$i.log 2>&1
done
python3 - $out "$kern" <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(float); n=collections.defaultdict(int)
for f in glob.glob(sys.argv[1]+"/p*/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            acc[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
w=acc.get("SQ_WAVES",0) or 1
print("kernel", sys.argv[2], "dispatches", max(n.values()) if n else 0, "waves", w)
for k,v in sorted(acc.items()): print("  %-28s total %.4g  per wave %.2f" % (k, v, v/w))
PY
