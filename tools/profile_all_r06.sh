#!/bin/bash
# the round's evidence for the workloads of the bench line + the rebuild timelines, one box:
#   gpurun --timeout 3000 -- 'bash tools/profile_all_r06.sh'
set -u
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
bash tools/profile_r06.sh 4m --steps 100 --warmup 20 > gpurun_out/prof_4m.txt 2>&1
bash tools/profile_r06.sh 1m --lattice 64 --steps 100 --warmup 20 > gpurun_out/prof_1m.txt 2>&1
bash tools/profile_r06.sh lipid --workload lipid --steps 100 --warmup 20 > gpurun_out/prof_lipid.txt 2>&1
bash tools/profile_r06.sh lipid40 --workload lipid --types 40 --steps 100 --warmup 20 > gpurun_out/prof_lipid40.txt 2>&1
bash tools/profile_r06.sh brick --lattice 51 --rccl-loopback --steps 100 --warmup 20 > gpurun_out/prof_brick.txt 2>&1
bash tools/profile_r06.sh lipidbrick --workload lipid --reps 6,6,3 --rccl-loopback --steps 100 --warmup 20 > gpurun_out/prof_lipidbrick.txt 2>&1
bash tools/timeline.sh 4m --steps 40 --warmup 20 --no-pmc > /dev/null 2>&1
bash tools/timeline.sh lipid --workload lipid --steps 40 --warmup 20 --no-pmc > /dev/null 2>&1
bash tools/timeline.sh brick --lattice 51 --rccl-loopback --steps 40 --warmup 20 --no-pmc > /dev/null 2>&1
bash tools/timeline.sh lipidbrick --workload lipid --reps 6,6,3 --rccl-loopback --steps 40 --warmup 20 --no-pmc > /dev/null 2>&1
bash tools/timeline.sh small --lattice 12 --steps 40 --warmup 20 --no-pmc > /dev/null 2>&1
for t in 4m 1m lipid lipid40 brick lipidbrick; do python3 - $t <<'PY'
import json, sys
t = sys.argv[1]
d = json.load(open("gpurun_out/prof_%s/traffic.json" % t))
print(t, d.get("bench_plain"), d.get("traffic_kernel"), d.get("traffic_bytes_per_launch"))
PY
done
cat gpurun_out/prof_*/warnings.txt 2>/dev/null
head -3 gpurun_out/timeline_*/timeline.txt
