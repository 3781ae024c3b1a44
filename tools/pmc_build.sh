#!/bin/bash
# SQ counters of the list-rebuild kernels (k_tile_build, k_tile_transpose): bash tools/pmc_build.sh [bench args]
set -u
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
export TMPDIR=/tmp
root=$PWD
out=gpurun_out/pmc_build
rm -rf $out; mkdir -p $out
args="${1:---lattice 100 --steps 4 --warmup 2 --equil 0}"
cd /tmp
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH"; do
   name=$(echo $set | tr ' ' '+' | cut -c1-30)
   timeout 300 rocprofv3 --pmc $set --output-format csv -d $root/$out/pmc_$name -o p -- python3 $root/bench.py $args --no-cpu > $root/$out/log_$name.txt 2>&1
done
cd $root
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for fn in glob.glob('gpurun_out/pmc_build/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r['Kernel_Name'].split('(')[0][:28]
        if 'k_tile' not in k and 'k_nonbond' not in k: continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[k][r['Counter_Name']] += 1
for k, v in acc.items():
    print("==", k)
    calls = max(n[k].values())
    for c in sorted(v): print("   %-26s %.4e per launch" % (c, v[c] / n[k][c]))
    wc = v.get('SQ_WAVE_CYCLES', 0) / max(n[k].get('SQ_WAVE_CYCLES', 1), 1)
    if wc:
        g = lambda c: v.get(c, 0) / max(n[k].get(c, 1), 1)
        print("   per wave-cycle: VALU active %.3f  LDS active %.3f  SALU/scalar active %.3f  wait_inst_any %.3f  wait_lds %.3f ; lane util %.3f ; VALU insts per wave %.0f, SALU %.0f, LDS %.0f, branch %.0f" % (
            g('SQ_ACTIVE_INST_VALU') / wc * 4, g('SQ_ACTIVE_INST_LDS') / wc * 4, g('SQ_ACTIVE_INST_SCA') / wc * 4, g('SQ_WAIT_INST_ANY') / wc * 4, g('SQ_WAIT_INST_LDS') / wc * 4,
            g('SQ_THREAD_CYCLES_VALU') / max(g('SQ_ACTIVE_INST_VALU') * 64, 1), g('SQ_INSTS_VALU') / max(g('SQ_WAVES'), 1), g('SQ_INSTS_SALU') / max(g('SQ_WAVES'), 1), g('SQ_INSTS_LDS') / max(g('SQ_WAVES'), 1), g('SQ_INSTS_BRANCH') / max(g('SQ_WAVES'), 1)))
PY
