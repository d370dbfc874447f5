#!/bin/bash
# Per-kernel HBM traffic of the bench command from rocprofv3 PMC counters, one counter per pass
# (MI355X_MICROARCH.md, "HBM" and "rocprofv3 PMC slots": FETCH_SIZE and WRITE_SIZE do not fit one pass).
# usage: tools/pmc_traffic.sh <outdir> -- <command...>
set -e
out=$1; shift; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d "$out/pmc_$c" --output-format csv -- "$@" > "$out/pmc_$c.log" 2>&1
done
