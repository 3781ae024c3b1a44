#!/bin/bash
# A/B of the list build on ONE box: tree vs every tuning/libddcmi_*.so -- wall time of the rebuild (water 4 M, 500 k; lipid 2 M), then the
# kernel-trace stats of the rebuild kernels.   gpurun --timeout 900 -- 'bash tools/ab_build_r04.sh'
set -u
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
for lat in 100 50; do
   for round in 1 2; do
      python3 tools/time_rebuild.py $lat 10
      for so in tuning/libddcmi_*.so; do [ -e "$so" ] || continue; DDCMI_LIB=$PWD/$so python3 tools/time_rebuild.py $lat 10; done
   done
done
WORKLOAD=lipid python3 tools/time_rebuild.py 0 10
for so in tuning/libddcmi_*.so; do [ -e "$so" ] || continue; WORKLOAD=lipid DDCMI_LIB=$PWD/$so python3 tools/time_rebuild.py 0 10; done
bash tools/prof_variants.sh "--no-also --lattice 100 --steps 40 --warmup 0 --equil 0"
