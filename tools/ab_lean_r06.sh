for rep in 1 2; do
for v in "" "DDCMI_LEAN_MAX_BEADS=100000000 DDCMI_LEAN_BONDED=1"; do
  for args in "--steps 100" "--workload lipid --steps 100"; do
    env $v python3 bench.py --no-cpu --no-also --no-pmc $args 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('%-55s %-28s step %.4f kernel %.4f frac %.3f' % ('$v'[:55], '$args', d['ms_per_step'], r['kernel_ms_avg'], r['frac']))
"
  done
done
done
