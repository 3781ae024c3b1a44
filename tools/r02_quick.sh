#!/bin/bash
# quick GPU check of a kernel change: parity subset, then the bench lines that matter
cd $GRAFT_REPO_ROOT
o=gpurun_out/r02_quick; mkdir -p $o
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_domains.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -4
b() { python3 bench.py --no-cpu "$@" 2>&1 | grep '^{' | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d['roofline']
    print('%-34s %8.4f ms/step  nonbond %7.1f us  frac %.3f  L %.1f  rebuilds %d' % (d['config']['workload'] + (' lb' if 'loopback' in d['config']['parallelism'] else ''), d['ms_per_step'], r['kernel_ms_avg'] * 1e3, r['frac'], d['config']['list_entries_per_atom'], d['config']['rebuilds_in_timed_region']))
"; }
b --steps 100 --warmup 20
b --lattice 64 --steps 200 --warmup 40
b --lattice 50 --steps 400 --warmup 40
b --lattice 50 --steps 400 --warmup 40 --rccl-loopback
b --workload lipid --steps 100 --warmup 20
