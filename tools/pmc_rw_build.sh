#!/bin/bash
# HBM bytes of the list-rebuild kernels (FETCH_SIZE, WRITE_SIZE: separate --pmc passes): bash tools/pmc_rw_build.sh [bench args]
set -u
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
export TMPDIR=/tmp
root=$PWD
out=gpurun_out/pmc_rw_build
rm -rf $out; mkdir -p $out
args="${1:---lattice 100 --steps 4 --warmup 2 --equil 0 --no-also}"
cd /tmp
for set in FETCH_SIZE WRITE_SIZE; do
   timeout 300 rocprofv3 --pmc $set --output-format csv -d $root/$out/pmc_$set -o p -- python3 $root/bench.py $args --no-cpu > $root/$out/log_$set.txt 2>&1
done
cd $root
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob('gpurun_out/pmc_rw_build/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r['Kernel_Name'].split('(')[0][:40]
        if 'k_tile' not in k: continue
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    print("==", k, {c: "%.1f MB per launch (KiB counter, as is)" % (sum(x) / len(x) * 1024 / 1e6) for c, x in v.items()})
PY
