#!/bin/bash
# A/B of an environment switch on one box: bash tools/r02_env_ab.sh VAR=1 "lb 500k 4m"
cd $GRAFT_REPO_ROOT
sw="$1"; only="${2:-lb 500k lip 4m 1m}"
b() { python3 bench.py --no-cpu "$@" 2>&1 | grep '^{' | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d['roofline']
    print('   %-34s %8.4f ms/step  nonbond %7.1f us  epot %.10g' % (d['config']['workload'] + (' lb' if 'loopback' in d['config']['parallelism'] else ''), d['ms_per_step'], r['kernel_ms_avg'] * 1e3, d['check']['epot']))
"; }
for rep in 1 2; do for mode in off on; do
   echo "== $sw $mode (rep $rep)"
   for w in $only; do
      case $w in 4m) a="--steps 100 --warmup 20";; 1m) a="--lattice 64 --steps 200 --warmup 40";; 500k) a="--lattice 50 --steps 400 --warmup 40";; lb) a="--lattice 50 --steps 400 --warmup 40 --rccl-loopback";; lip) a="--workload lipid --steps 100 --warmup 20";; esac
      if [ $mode = on ]; then env $sw python3 -c "pass"; (export $sw; b $a); else b $a; fi
   done
done; done
