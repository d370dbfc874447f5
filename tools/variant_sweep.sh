#!/bin/bash
# Throughput of the fused-RK4 kernel variants over grid sizes (on the GPU box): picks the defaults in
# marl_api.hip:default_rk4_variant.   usage: tools/variant_sweep.sh "16384 65536" "2 3 4"   (variant = index into kRk4Variants: 1, 2, 4, 8, 16 steps per launch)
for n in $1; do for v in $2; do
  echo -n "N=$n variant $v: "
  python bench.py --no-cpu-baseline --n $n --variant $v 2>/dev/null | python -c "import sys,json; print('%.3e' % json.loads(sys.stdin.read())['value'])"
done; done
