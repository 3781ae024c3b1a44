#!/bin/bash
cd $GRAFT_REPO_ROOT
b() { DDCMI_DEBUG_SCHED=$dbg python3 bench.py --no-cpu "$@" 2>&1 | grep -E '^{|ddcmi sched: xcd 0' | sort -u | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('ddcmi'): print('      ', l.strip()); continue
    d = json.loads(l); r = d['roofline']
    print('   %-34s %8.4f ms/step  nonbond %7.1f us frac %.3f' % (d['config']['workload'] + (' lb' if 'loopback' in d['config']['parallelism'] else ''), d['ms_per_step'], r['kernel_ms_avg'] * 1e3, r['frac']))
"; }
dbg=1
for rep in 1 2; do
  b --lattice 50 --steps 400 --warmup 40
  b --lattice 64 --steps 200 --warmup 40
  b --lattice 32 --steps 400 --warmup 40
  dbg=
done
