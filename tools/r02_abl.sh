#!/bin/bash
cd $GRAFT_REPO_ROOT
n=${1:-100}; shift
for rep in 1 2; do
for name in "$@"; do
   if [ "$name" = tree ]; then DDCMI_LIB=$PWD/ddcmd_amd/lib/libddcmi.so python3 tools/time_nonbond.py $n; else DDCMI_LIB=$PWD/ddcmd_amd/lib/variants/libddcmi_$name.so python3 tools/time_nonbond.py $n; fi
done; done 2>&1 | grep k_nonbond
