#!/bin/bash
cd $GRAFT_REPO_ROOT
export DDCMI_LIB=$PWD/ddcmd_amd/lib/variants/libddcmi_trace.so
for n in 50 64 100; do echo "== n=$n"; DDCMI_DEBUG_SCHED=1 python3 tools/trace_gaps.py $n 2>&1 | tail -16; done
