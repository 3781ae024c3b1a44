#!/bin/bash
cd $GRAFT_REPO_ROOT
b() { python3 bench.py --no-cpu "$@" 2>&1 | grep '^{' | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d['roofline']
    print('   %-34s %8.4f ms/step  nonbond %7.1f us' % (d['config']['workload'], d['ms_per_step'], r['kernel_ms_avg'] * 1e3))
"; }
for rep in 1 2; do
for ov in 0 1; do
  echo "== overlap $ov"
  DDCMI_HALO_OVERLAP=$ov b --lattice 50 --steps 400 --warmup 40 --rccl-loopback
done; done
