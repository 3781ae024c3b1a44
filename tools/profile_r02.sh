#!/bin/bash
# Round-2 evidence for bench.py's roofline block, per workload.  Run on the GPU box from the repo root:
#   bash tools/profile_r02.sh <tag> <bench args...>      e.g.  bash tools/profile_r02.sh 4m --steps 100 --warmup 20
# Kernel-trace stats and each PMC set are SEPARATE rocprofv3 runs (never combined); outputs under gpurun_out/prof_r02_<tag>,
# tools/curate_r02.py turns them into profiles/r02_<tag>_{kernel_stats.csv,traffic.json,bench.log}.
tag=$1; shift
export TMPDIR=/tmp
root=$PWD
out=$root/gpurun_out/prof_r02_$tag
rm -rf $out; mkdir -p $out
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python3 $root/bench.py --no-cpu "$@" > $out/bench_stats.log 2>&1
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INST_CYCLES_VMEM"; do
   name=$(echo $set | tr ' ' '+' | cut -c1-40)
   pmcargs=$(echo "$@" | sed -E 's/--steps [0-9]+/--steps 4/; s/--warmup [0-9]+/--warmup 2/')
   timeout 900 rocprofv3 --pmc $set --output-format csv -d $out/pmc_$name -o p -- python3 $root/bench.py --no-cpu $pmcargs > $out/bench_pmc_$name.log 2>&1
done
cd $root
# the plain (un-profiled) bench line of the same command, for the record
python3 bench.py --no-cpu "$@" > $out/bench_plain.log 2>&1
python3 tools/summarize_profile.py $out > $out/summary.json
python3 - <<PY
import json
d = json.load(open("$out/summary.json"))
print("$tag:", [(k["name"][:28], round(k["avg_us"], 1), k["calls"]) for k in d["kernel_stats"][:6]])
p = d["pmc_k_nonbond_mean_per_launch"]
print("   FETCH_SIZE KiB %.0f  WRITE_SIZE KiB %.0f  -> traffic %.4e B/launch" % (p.get("FETCH_SIZE", 0), p.get("WRITE_SIZE", 0), (2 * p.get("FETCH_SIZE", 0) + p.get("WRITE_SIZE", 0)) * 1024))
PY
grep '^{' $out/bench_plain.log | cut -c1-200
