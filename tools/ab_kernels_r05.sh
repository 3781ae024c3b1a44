#!/bin/bash
# per-kernel average durations (rocprofv3 --kernel-trace --stats) of one bench invocation, the tree against tuning/libddcmi_*.so, one box:
#   bash tools/ab_kernels_r05.sh <kernel name pattern> "<bench args>" [rounds]
set -u
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
pat=$1; args=$2; rounds=${3:-2}
export TMPDIR=/tmp
root=$PWD
for r in $(seq 1 $rounds); do
   for so in tree tuning/libddcmi_*.so; do
      if [ "$so" = tree ]; then unset DDCMI_LIB; else [ -e "$so" ] || continue; export DDCMI_LIB=$root/$so; fi
      d=/tmp/abk_$$; rm -rf $d
      (cd /tmp && timeout ${ABK_TIMEOUT:-200} rocprofv3 --kernel-trace --stats --output-format csv -d $d -o s -- python3 $root/bench.py --no-cpu --no-also --no-pmc $args > $d.log 2>&1)
      python3 - "$d" "$pat" "$(basename $so)" "$d.log" <<'PY'
import csv, glob, json, os, sys
d, pat, name, log = sys.argv[1:5]
ms = "?"
for line in open(log):
    if line.startswith("{"): ms = "%.4f" % json.loads(line)["ms_per_step"]
for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Name"]:
            print("%-28s ms/step(profiled) %s   %-60s calls %5s avg %9.2f us" % (name, ms, r["Name"].split("(")[0][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
      rm -rf $d $d.log
   done
done
unset DDCMI_LIB
