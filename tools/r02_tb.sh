#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in ${VARIANTS:-trace}; do
export DDCMI_LIB=$PWD/ddcmd_amd/lib/variants/libddcmi_$v.so
for a in ${SIZES:-50 64 100}; do echo "== $v $a"; timeout 300 python3 tools/trace_build.py $a 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl" | grep "${FILTER:-.}" | tail -12; done; done
