#!/bin/bash
# A/B of tuning builds on ONE box: wall time of the list rebuild (tools/time_rebuild.py) for every tuning/libddcmi_*.so and the tree's library.
#   gpurun --timeout 600 -- 'bash tools/ab_rebuild.sh [lattice] [reps]'
set -u
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
lat=${1:-100}; reps=${2:-10}
for round in 1 2; do
   python3 tools/time_rebuild.py $lat $reps
   for so in tuning/libddcmi_*.so; do
      [ -e "$so" ] || continue
      DDCMI_LIB=$PWD/$so python3 tools/time_rebuild.py $lat $reps
   done
done
if [ "${3:-}" = "lipid" ]; then
   WORKLOAD=lipid python3 tools/time_rebuild.py $lat $reps
   for so in tuning/libddcmi_*.so; do
      [ -e "$so" ] || continue
      WORKLOAD=lipid DDCMI_LIB=$PWD/$so python3 tools/time_rebuild.py $lat $reps
   done
fi
