#!/bin/bash
# A/B of two library builds on one box:  tools/ab_bench.sh "<bench args>" libA.so libB.so [rounds]
args=$1; A=$2; B=$3; rounds=${4:-3}
for r in $(seq $rounds); do for lib in $A $B; do
  v=$(MARL_HIP_LIBRARY=$PWD/$lib python3 bench.py --no-cpu-baseline --no-extras $args 2>/dev/null | python3 -c "import sys,json; print('%.4e' % json.loads(sys.stdin.read())['value'])")
  echo "$(basename $lib) [$args] $v"
done; done
