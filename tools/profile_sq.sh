#!/bin/bash
# SQ counter passes (instruction mix, wave cycles, waits) of one bench workload:  tools/profile_sq.sh <outdir> <tag> <bench args...>
set -e
out=$1; tag=$2; shift 2
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d "$out/pmc_${tag}_SQ1" --output-format csv -- python3 bench.py --no-cpu-baseline --no-extras "$@" > "$out/pmc_${tag}_SQ1.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY -d "$out/pmc_${tag}_SQ2" --output-format csv -- python3 bench.py --no-cpu-baseline --no-extras "$@" > "$out/pmc_${tag}_SQ2.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d "$out/pmc_${tag}_SQ3" --output-format csv -- python3 bench.py --no-cpu-baseline --no-extras "$@" > "$out/pmc_${tag}_SQ3.log" 2>&1
echo "sq $tag done"
