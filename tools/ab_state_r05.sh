#!/bin/bash
# the plain pair kernel of the tree and of every tuning/libddcmi_*.so on ONE saved state, R rounds:   bash tools/ab_state_r05.sh <lattice> <rounds>
set -u
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
n=${1:-100}; rounds=${2:-2}
unset DDCMI_LIB
for r in $(seq 1 $rounds); do
   python3 tools/time_nonbond_state.py $n 30 2>/dev/null | tail -1
   for so in tuning/libddcmi_*.so; do [ -e "$so" ] || continue; DDCMI_LIB=$PWD/$so python3 tools/time_nonbond_state.py $n 30 2>/dev/null | tail -1; done
done
