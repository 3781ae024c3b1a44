#!/bin/bash
# A/B of ONE environment switch through bench.py on one box, N rounds:   bash tools/ab_env_r05.sh VAR ROUNDS "<bench args>" ["<bench args 2>" ...]
set -u
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
var=$1; rounds=$2; shift 2
for args in "$@"; do
   echo "### $args"
   for r in $(seq 1 $rounds); do
      for e in 0 1; do
         if [ $e = 1 ]; then export $var=1; else unset $var; fi
         python3 bench.py --no-cpu --no-also --no-pmc $args 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('  %s=%s  ms/step %.4f  kernel %.4f ms  windows %s' % ('$var', '$e', d['ms_per_step'], r['kernel_ms_avg'], d['window_ms']))"
      done
   done
done
unset $var
