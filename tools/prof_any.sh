#!/bin/bash
# kernel-trace stats of an arbitrary bench invocation: bash tools/prof_any.sh <tag> <bench args...>
set -u
tag=$1; shift
export TMPDIR=/tmp
root=$PWD
out=gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/stats -o s -- python3 $root/bench.py --no-cpu "$@" > $root/$out/bench_stats.log 2>&1
cd $root
grep '^{' $out/bench_stats.log | cut -c1-330
python3 - <<PY
import csv, glob
for f in glob.glob("$out/stats/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("total kernel ms", tot/1e6)
    for r in rows[:14]:
        print("%-60s calls %6s avg_us %10.2f total_ms %9.3f pct %6s" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6, r["Percentage"]))
PY
