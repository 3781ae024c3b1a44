#!/bin/bash
cd $GRAFT_REPO_ROOT
n=${1:-100}; shift
for rep in 1 2; do
for name in "$@"; do
   if [ "$name" = tree ]; then lib=$PWD/ddcmd_amd/lib/libddcmi.so; else lib=$PWD/ddcmd_amd/lib/variants/libddcmi_$name.so; fi
   DDCMI_LIB=$lib python3 tools/time_rebuild.py $n
done; done 2>&1 | grep rebuild
