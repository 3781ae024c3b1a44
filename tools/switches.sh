#!/bin/bash
# the documented environment switches against the oracle: the water/decomposition parity tests under each of them
#   gpurun --timeout 1500 -- 'bash tools/switches.sh'
set -u
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
for sw in "" DDCMI_NO_FUSED_STEP=1 DDCMI_NO_LEAN_STEP=1 DDCMI_NO_SELF_IMAGES=1 DDCMI_NO_BONDED_LDS_TABLES=1 DDCMI_NO_SHELL_SKIP=1 DDCMI_HALO_OVERLAP=1 DDCMI_DEBUG_GUARD=1; do
   echo "== ${sw:-defaults}"
   env $sw timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_domains.py tests/test_gpu_rccl_loopback.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -2
done
