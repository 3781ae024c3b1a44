#!/bin/bash
# kernel-trace stats of bench invocations for the libraries given: bash tools/r02_prof.sh "<bench args>" tree base ...
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
args="$1"; shift
root=$PWD
for name in "$@"; do
   if [ "$name" = tree ]; then unset DDCMI_LIB; else export DDCMI_LIB=$root/ddcmd_amd/lib/variants/libddcmi_$name.so; fi
   out=gpurun_out/r02_prof_$name; rm -rf $out; mkdir -p $out
   (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/stats -o s -- python3 $root/bench.py --no-cpu $args > $root/$out/bench.log 2>&1)
   echo "== $name: $(grep '^{' $out/bench.log | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.4f ms/step' % d['ms_per_step'])")"
   python3 - <<PY
import csv, glob
for f in glob.glob("$out/stats/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("total kernel ms %.2f" % (tot/1e6))
    for r in rows[:22]:
        print("  %-52s calls %5s avg_us %9.2f total_ms %8.3f" % (r["Name"][:52], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
done
