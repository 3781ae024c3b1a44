# VALU lane utilisation of the kernels: SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU * 64).
# Run on the GPU box from the repo root: bash tools/pmc_util.sh
export TMPDIR=/tmp
root=$PWD
out=gpurun_out/pmc_util
rm -rf $out; mkdir -p $out
cd /tmp
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d $root/$out/pmc -o p -- python3 $root/bench.py --lattice 100 --steps 4 --warmup 2 --no-cpu > $root/$out/log.txt 2>&1
cd $root
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/pmc_util/pmc/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        k = r['Kernel_Name'][:40]
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0))[:6]:
    t, a = v.get('SQ_THREAD_CYCLES_VALU', 0), v.get('SQ_ACTIVE_INST_VALU', 0)
    print("%-42s thread_cycles %.3e active_inst %.3e insts %.3e  util %.3f" % (k, t, a, v.get('SQ_INSTS_VALU', 0), t / (a * 64) if a else 0))
PY
