#!/bin/bash
# a longer randomised hunt than the test suite's (new seeds): bash tools/r02_fuzz.sh
cd $GRAFT_REPO_ROOT
f() { echo "== $*"; timeout 900 python3 "$@" 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl" | tail -3; }
f tools/fuzz_parity.py 80 20261002
f tools/fuzz_parity.py 40 777 domains
f tools/fuzz_parity.py 40 778 loopback
f tools/fuzz_lipid.py 24 4242
f tools/fuzz_features.py 30 99
f tools/fuzz_reuse.py 30 5
f tools/long_run_domains.py 1500
