#!/bin/bash
# One GPU call: the -m gpu tests, then kernel-trace stats of the headline and the lipid workload.
#   gpurun --timeout 900 -- 'bash tools/gpu_check.sh <tag> [notest]'
set -u
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
tag=${1:-x}
mkdir -p gpurun_out
if [ "${2:-}" != "notest" ]; then
   timeout 600 python3 -m pytest tests -m gpu -x -q > gpurun_out/${tag}_pytest.log 2>&1
   tail -5 gpurun_out/${tag}_pytest.log
fi
bash tools/prof_any.sh ${tag}_4m --no-also --lattice 100 --steps 60 --warmup 20 2>&1 | cut -c1-200
bash tools/prof_any.sh ${tag}_lipid --no-also --workload lipid --steps 60 --warmup 20 2>&1 | cut -c1-200
