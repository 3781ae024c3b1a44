#!/bin/bash
# kernel-trace stats of the rebuild kernels for the tree's library and every tuning build, one box
set -u
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
args="${1:---no-also --lattice 100 --steps 40 --warmup 0 --equil 0}"
for so in tree tuning/libddcmi_*.so; do
   [ "$so" = tree ] || [ -e "$so" ] || continue
   if [ "$so" = tree ]; then unset DDCMI_LIB; else export DDCMI_LIB=$PWD/$so; fi
   echo "== $so"
   bash tools/prof_any.sh v_$(basename $so .so) $args 2>&1 | grep -E "k_tile|k_nonbond|k_gather_state|ms_per_step" | cut -c1-150
done
