#!/bin/bash
# the documented environment switches still produce oracle-exact results: bash tools/r02_switches.sh
cd $GRAFT_REPO_ROOT
f() { echo "== $*"; env "$@" 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl" | tail -1; }
f DDCMI_HALO_OVERLAP=1 timeout 600 python3 tools/fuzz_parity.py 20 31 loopback
f DDCMI_NO_LPT=1 timeout 600 python3 tools/fuzz_parity.py 20 32
f DDCMI_NO_TAIL_SPLIT=1 timeout 600 python3 tools/fuzz_parity.py 20 33
f DDCMI_LPT_ROUNDS=100 timeout 600 python3 tools/fuzz_parity.py 20 34 domains
f DDCMI_GRAPH_MAX_BEADS=100000 timeout 600 python3 tools/fuzz_parity.py 20 35
f DDCMI_DEBUG_GUARD=1 timeout 600 python3 tools/fuzz_lipid.py 8 36
