#!/bin/bash
# A/B of library builds on ONE box: tools/lib_ab.sh "<bench args>" "<MARL_HIP_OPTIONS>" rounds lib1.so lib2.so ...
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"
args=$1; opts=$2; rounds=$3; shift 3
for r in $(seq $rounds); do for lib in "$@"; do
  v=$(MARL_HIP_LIBRARY=$ROOT/$lib MARL_HIP_OPTIONS="$opts" python3 bench.py --no-cpu-baseline --no-extras $args 2>/tmp/ab_err.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4e  ms/step %.5f' % (d['value'], d['ms_per_step']))" || tail -3 /tmp/ab_err.txt)
  echo "$(basename $lib) [$opts] [$args] $v"
done; done
