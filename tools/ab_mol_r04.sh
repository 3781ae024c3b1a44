#!/bin/bash
# lipid list build: tree vs tuning builds (wall time of the rebuild, then the kernels).   gpurun --timeout 900 -- 'bash tools/ab_mol_r04.sh'
set -u
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_domains.py tests/test_gpu_driver.py -m gpu -x -q 2>&1 | grep -i "passed\|failed\|error" | tail -4
for r in 1 2; do
WORKLOAD=lipid python3 tools/time_rebuild.py 0 10
for so in tuning/libddcmi_*.so; do [ -e "$so" ] || continue; WORKLOAD=lipid DDCMI_LIB=$PWD/$so python3 tools/time_rebuild.py 0 10; done
done
