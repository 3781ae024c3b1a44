#!/bin/bash
# A/B of one environment switch on ONE box through bench.py:   bash tools/ab_env_r04.sh DDCMI_NO_TRUE_DISPLACEMENT "<bench args>" ["<bench args 2>" ...]
set -u
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
var=$1; shift
run() { python3 bench.py --no-cpu --no-also --no-pmc "$@" 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('   ms/step %.4f  kernel %.4f ms (%s)' % (d['ms_per_step'], r['kernel_ms_avg'], r['kernel'][:16]))
"; }
for args in "$@"; do
   echo "### $args"
   for r in 1 2; do
      echo " default"; run $args
      echo " $var=1"; export $var=1; run $args; unset $var
   done
done
