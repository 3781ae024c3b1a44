#!/bin/bash
# the lean step and the images staged from their owners against the reduction + image launch per step, one box:   bash tools/ab_lean_r05.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { python3 bench.py --no-cpu --no-also --no-pmc $2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('  %-40s ms/step %.4f  kernel %.4f ms' % ('$1', d['ms_per_step'], r['kernel_ms_avg']))"; }
for args in "--steps 200" "--lattice 64 --steps 400" "--workload lipid --steps 100"; do
 echo "### $args"
 for r in 1 2; do
  ( export DDCMI_LEAN_MAX_BEADS=100000000; run "lean+self" "$args" )
  ( export DDCMI_NO_LEAN_STEP=1; run "self images, reduce per step" "$args" )
  ( export DDCMI_NO_LEAN_STEP=1 DDCMI_NO_SELF_IMAGES=1; run "old" "$args" )
 done
done
