#!/bin/bash
# one list rebuild's timeline (kernels, copies, idle gaps):  gpurun -- 'bash tools/timeline.sh <tag> <bench args...>'
set -u
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
tag=$1; shift
export TMPDIR=/tmp
root=$PWD
out="gpurun_out/timeline_$tag"; rm -rf "$out"; mkdir -p "$out"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$root/$out/t" -o s -- python3 "$root/bench.py" --no-cpu --no-also "$@" > "$root/$out/bench.log" 2>&1)
grep '^{' "$out/bench.log" | cut -c1-200
python3 tools/rebuild_timeline.py "$out/t" > "$out/timeline.txt"; cat "$out/timeline.txt"
rm -rf "$out/t"
