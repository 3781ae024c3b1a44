#!/bin/bash
cd $GRAFT_REPO_ROOT
b() { python3 bench.py --no-cpu "$@" 2>&1 | grep '^{' | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d['roofline']
    print('   %-34s %8.4f ms/step  nonbond %7.1f us  frac %.3f' % (d['config']['workload'], d['ms_per_step'], r['kernel_ms_avg'] * 1e3, r['frac']))
"; }
for rep in 1 2; do
echo split; b --lattice 50 --steps 400 --warmup 40
echo nosplit; DDCMI_NO_TAIL_SPLIT=1 b --lattice 50 --steps 400 --warmup 40
echo base; DDCMI_LIB=$PWD/ddcmd_amd/lib/variants/libddcmi_base.so b --lattice 50 --steps 400 --warmup 40
done
