#!/bin/bash
# A/B of tuning builds on ONE box through bench.py: tree, then every tuning/libddcmi_*.so.   bash tools/ab_bench_r04.sh "<bench args>" ["<bench args 2>" ...]
set -u
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
for args in "$@"; do
   echo "### $args"
   for so in tree tuning/libddcmi_*.so; do
      [ "$so" = tree ] || [ -e "$so" ] || continue
      if [ "$so" = tree ]; then unset DDCMI_LIB; else export DDCMI_LIB=$PWD/$so; fi
      python3 bench.py --no-cpu --no-also $args 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; o=r.get('other_launches') or {}
        print('%-28s ms/step %.4f  kernel %.4f ms (%s) frac %.3f  other %.4f' % ('$so'.split('/')[-1], d['ms_per_step'], r['kernel_ms_avg'], r['kernel'][:16], r['frac'], o.get('kernel_ms_avg',0)))
"
   done
done
