#!/bin/bash
# One profiling pass of the round's final state (run on the GPU box through gpurun):
#   tools/profile_round.sh gpurun_out/r01z
# Writes bench JSON lines of every workload, the rocprofv3 kernel-trace statistics of the default bench and the
# PMC passes (HBM traffic: FETCH_SIZE / WRITE_SIZE, one counter per pass; SQ instruction counters) under <outdir>.
# tools/summarize_profiles.py turns them into the files committed under profiles/.
set -e
out=$1
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python bench.py > "$out/bench_default.json" 2> "$out/bench_default.err"
echo "bench default done"
for w in sweep_rk45 rk45_single sweep_rk4 dd_rk45; do
  python bench.py --no-cpu-baseline --workload $w > "$out/bench_$w.json" 2> "$out/bench_$w.err"
done
python bench.py --no-cpu-baseline --n 65536 > "$out/bench_n65536.json" 2> "$out/bench_n65536.err"
echo "benches done"
rocprofv3 --kernel-trace --stats -d "$out/stats" --output-format csv -- python bench.py --no-cpu-baseline > "$out/stats.log" 2>&1
echo "kernel stats done"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d "$out/pmc_$c" --output-format csv -- python bench.py --no-cpu-baseline --steps 400 > "$out/pmc_$c.log" 2>&1
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d "$out/pmc_SQ1" --output-format csv -- python bench.py --no-cpu-baseline --steps 400 > "$out/pmc_SQ1.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY -d "$out/pmc_SQ2" --output-format csv -- python bench.py --no-cpu-baseline --steps 400 > "$out/pmc_SQ2.log" 2>&1
echo "pmc done"
