#!/bin/bash
# One profiling pass of the round's state (run on the GPU box through gpurun):
#   tools/profile_round.sh gpurun_out/r02a [quick]
# Writes, under <outdir>: the bench JSON line of every workload; rocprofv3 kernel-trace statistics of every workload
# (stats_<workload>/); PMC passes of the dominant kernels (HBM traffic: FETCH_SIZE / WRITE_SIZE, one counter per pass;
# SQ instruction / wait counters) in pmc_<workload>_<set>/.  tools/summarize_profiles.py turns them into the files
# committed under profiles/.  Counter passes never share a run with a trace (gpurun refuses that combination).
set -e
out=$1
quick=$2
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 bench.py > "$out/bench_default.json" 2> "$out/bench_default.err"
echo "bench default done"
for w in sweep_rk45 rk45_single sweep_rk4 dd_rk45; do
  python3 bench.py --no-cpu-baseline --workload $w > "$out/bench_$w.json" 2> "$out/bench_$w.err"
done
python3 bench.py --no-cpu-baseline --no-extras --n 65536 > "$out/bench_n65536.json" 2> "$out/bench_n65536.err"
echo "benches done"
[ "$quick" = "quick" ] && exit 0
# kernel-trace statistics per workload (short runs: the trace of every dispatch is kept)
# (the bench line of every TRACED run is kept beside the untraced one - traced_<workload>.json: `ms_per_step` of the run whose kernel times
# the statistics table holds; tools/summarize_profiles.py puts the two side by side)
rocprofv3 --kernel-trace --stats -d "$out/stats_default" --output-format csv -- python3 bench.py --no-cpu-baseline --no-extras > "$out/traced_default.json" 2> "$out/stats_default.log"
rocprofv3 --kernel-trace --stats -d "$out/stats_n65536" --output-format csv -- python3 bench.py --no-cpu-baseline --no-extras --n 65536 > "$out/traced_n65536.json" 2> "$out/stats_n65536.log"
for w in sweep_rk45 rk45_single sweep_rk4 dd_rk45; do
  rocprofv3 --kernel-trace --stats -d "$out/stats_$w" --output-format csv -- python3 bench.py --no-cpu-baseline --workload $w --steps 400 > "$out/traced_$w.json" 2> "$out/stats_$w.log"
  python3 bench.py --no-cpu-baseline --workload $w --steps 400 > "$out/bench400_$w.json" 2> /dev/null   # the same command untraced
done
# the domain-decomposed loop at the 8-GPU shard size with a real ncclAllGather in the stream (one-rank communicator), and the persistent
# single-grid loop at a size where it is the default
MARL_DD_TRANSPORT=rccl1 rocprofv3 --kernel-trace --stats -d "$out/stats_dd_shard" --output-format csv -- python3 bench.py --no-cpu-baseline --workload dd_rk45 --n 524288 --steps 400 > "$out/traced_dd_shard.json" 2> "$out/stats_dd_shard.log"
MARL_DD_TRANSPORT=rccl1 python3 bench.py --no-cpu-baseline --workload dd_rk45 --n 524288 --steps 400 > "$out/bench400_dd_shard.json" 2> /dev/null
rocprofv3 --kernel-trace --stats -d "$out/stats_rk45_n524288" --output-format csv -- python3 bench.py --no-cpu-baseline --workload rk45_single --n 524288 --steps 400 > "$out/traced_rk45_n524288.json" 2> "$out/stats_rk45_n524288.log"
python3 bench.py --no-cpu-baseline --workload rk45_single --n 524288 --steps 400 > "$out/bench400_rk45_n524288.json" 2> /dev/null
# the implicit path (the reference's DEFAULT solver): kernel statistics of single Radau runs (N = 200 / 16 000 / 64 000), BDF and a sweep;
# HBM counters of the cyclic-reduction kernels on the large grid (one counter per pass)
rocprofv3 --kernel-trace --stats -d "$out/stats_radau_single" --output-format csv -- python3 tools/radau_profile.py single > "$out/stats_radau_single.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$out/stats_radau_bdf_sweep" --output-format csv -- python3 tools/radau_profile.py bdf sweep > "$out/stats_radau_bdf_sweep.log" 2>&1
echo "kernel stats done"
SQ1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS"
SQ2="SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY"
pmc() {  # pmc <tag> <counters...> -- <bench args...>
  tag=$1; shift
  ctr=()
  while [ "$1" != "--" ]; do ctr+=("$1"); shift; done
  shift
  rocprofv3 --pmc "${ctr[@]}" -d "$out/pmc_$tag" --output-format csv -- python3 bench.py --no-cpu-baseline --no-extras "$@" > "$out/pmc_$tag.log" 2>&1
}
pmc default_FETCH_SIZE FETCH_SIZE -- --steps 400 --warmup 400
pmc default_WRITE_SIZE WRITE_SIZE -- --steps 400 --warmup 400
pmc default_SQ1 $SQ1 -- --steps 400 --warmup 400
pmc default_SQ2 $SQ2 -- --steps 400 --warmup 400
pmc default_no_reuse_SQ1 $SQ1 -- --steps 400 --warmup 400 --no-reuse
pmc rk45_single_FETCH_SIZE FETCH_SIZE -- --workload rk45_single --steps 200
pmc rk45_single_WRITE_SIZE WRITE_SIZE -- --workload rk45_single --steps 200
pmc rk45_single_SQ1 $SQ1 -- --workload rk45_single --steps 200
pmc rk45_single_SQ2 $SQ2 -- --workload rk45_single --steps 200
pmc sweep_rk45_SQ1 $SQ1 -- --workload sweep_rk45 --steps 200 --warmup 5
pmc sweep_rk45_SQ2 $SQ2 -- --workload sweep_rk45 --steps 200 --warmup 5
pmc sweep_rk4_SQ1 $SQ1 -- --workload sweep_rk4 --steps 200 --warmup 5
pmc n65536_SQ1 $SQ1 -- --n 65536 --steps 800
pmc n65536_SQ2 $SQ2 -- --n 65536 --steps 800
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d "$out/pmc_radau_single_$c" --output-format csv -- python3 tools/radau_profile.py single > "$out/pmc_radau_single_$c.log" 2>&1
done
rocprofv3 --pmc $SQ1 -d "$out/pmc_radau_single_SQ1" --output-format csv -- python3 tools/radau_profile.py single > "$out/pmc_radau_single_SQ1.log" 2>&1
echo "pmc done"
