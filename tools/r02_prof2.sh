#!/bin/bash
# kernel-trace stats of an arbitrary python tool: bash tools/r02_prof2.sh <name> <script> <args...>
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
root=$PWD
name=$1; shift
if [ "$name" = tree ]; then unset DDCMI_LIB; else export DDCMI_LIB=$root/ddcmd_amd/lib/variants/libddcmi_$name.so; fi
out=gpurun_out/r02_prof2_$name; rm -rf $out; mkdir -p $out
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/stats -o s -- python3 $root/"$@" > $root/$out/run.log 2>&1)
tail -2 $out/run.log
python3 - <<PY
import csv, glob
for f in glob.glob("$out/stats/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("total kernel ms %.2f" % (tot/1e6))
    for r in rows[:24]:
        print("  %-52s calls %5s avg_us %9.2f total_ms %8.3f" % (r["Name"][:52], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
