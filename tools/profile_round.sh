#!/bin/bash
# Collect the rocprofv3 evidence bench.py's roofline block refers to.  Run on the GPU
# box from the repo root:  bash tools/profile_round.sh r01b [lattice]   (lattice 100 = 4.0M beads, 64 = 1.05M)
# Kernel-trace stats and each PMC set are separate runs (never combined).
tag=${1:-r01}
lat=${2:-100}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
root=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/stats -o s -- python3 $root/bench.py --lattice $lat --steps 100 --warmup 40 --no-cpu > $root/$out/bench_stats.log 2>&1
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INST_CYCLES_VMEM"; do
   name=$(echo $set | tr ' ' '+' | cut -c1-40)
   rocprofv3 --pmc $set --output-format csv -d $root/$out/pmc_$name -o p -- python3 $root/bench.py --lattice $lat --steps 4 --warmup 2 --no-cpu > $root/$out/bench_pmc_$name.log 2>&1
done
cd $root
python3 tools/summarize_profile.py $out > $out/summary.json
cat $out/summary.json
