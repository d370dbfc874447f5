#!/bin/bash
# A/B of option sets on ONE box: tools/opt_ab.sh "<bench args>" "<opts A>" "<opts B>" [rounds]
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"
args=$1; A=$2; B=$3; rounds=${4:-3}
for r in $(seq $rounds); do for o in "$A" "$B"; do
  v=$(MARL_HIP_OPTIONS="$o" python3 bench.py --no-cpu-baseline --no-extras $args 2>/tmp/ab_err.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4e  ms/step %.5f' % (d['value'], d['ms_per_step']))" || tail -5 /tmp/ab_err.txt)
  echo "[$o] [$args] $v"
done; done
