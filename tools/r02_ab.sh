#!/bin/bash
# A/B on ONE box (devices differ by up to 12 %): the libraries named on the command line ("" = the tree's own
# libddcmi.so, else ddcmd_amd/lib/variants/libddcmi_<name>.so), each over the bench lines that matter.
#   bash tools/r02_ab.sh [--parity] [--only "4m 1m"] base "" other
cd $GRAFT_REPO_ROOT
parity=0; only="4m 1m 500k lb lip"
while [ $# -gt 0 ]; do case "$1" in --parity) parity=1; shift;; --only) only="$2"; shift 2;; *) break;; esac; done
o=gpurun_out/r02_ab; mkdir -p $o
if [ $parity = 1 ]; then python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_domains.py tests/test_gpu_fuzz.py tests/test_gpu_edge_fullsize.py -m gpu -x -q > $o/parity.log 2>&1; tail -3 $o/parity.log; fi
b() { python3 bench.py --no-cpu "$@" 2>&1 | grep '^{' | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d['roofline']
    print('   %-34s %8.4f ms/step  nonbond %7.1f us  frac %.3f  L %.1f  epot %.10g' % (d['config']['workload'] + (' lb' if 'loopback' in d['config']['parallelism'] else ''), d['ms_per_step'], r['kernel_ms_avg'] * 1e3, r['frac'], d['config']['list_entries_per_atom'], d['check']['epot']))
"; }
for rep in 1 2; do
for name in "$@"; do
   if [ -z "$name" ]; then unset DDCMI_LIB; echo "== tree (rep $rep)"; else export DDCMI_LIB=$PWD/ddcmd_amd/lib/variants/libddcmi_$name.so; echo "== $name (rep $rep)"; fi
   for w in $only; do case $w in
      4m) b --steps 100 --warmup 20;;
      1m) b --lattice 64 --steps 200 --warmup 40;;
      500k) b --lattice 50 --steps 400 --warmup 40;;
      lb) b --lattice 50 --steps 400 --warmup 40 --rccl-loopback;;
      lip) b --workload lipid --steps 100 --warmup 20;;
   esac; done
done; done
