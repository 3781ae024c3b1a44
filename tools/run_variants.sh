#!/bin/bash
# bench every tuning build under ddcmd_amd/lib/variants plus the default library
for lib in "" ddcmd_amd/lib/variants/*.so; do
   echo "== ${lib:-default}"
   DDCMI_LIB=${lib:+$PWD/$lib} python3 bench.py --lattice 100 --steps 60 --warmup 40 --no-cpu 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('ms/step %.4f  k_nonbond %.4f ms  epot %.9g' % (d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['check']['epot']))
    elif 'rror' in l: print(l.strip()[:300])
"
done
