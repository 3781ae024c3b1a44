#!/bin/bash
# bash tools/r02_timeline.sh "<bench args>"  -> one rebuild's timeline
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
root=$PWD
out=gpurun_out/r02_timeline; rm -rf $out; mkdir -p $out
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $root/$out/t -o s -- python3 $root/bench.py --no-cpu $1 > $root/$out/bench.log 2>&1)
grep '^{' $out/bench.log | cut -c1-200
python3 tools/rebuild_timeline.py $out/t > $out/timeline.txt; cat $out/timeline.txt
rm -rf $out/t
