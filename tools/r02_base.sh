#!/bin/bash
# round-2 baseline: numbers of the tree as round 1 left it
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
o=gpurun_out/r02_base; mkdir -p $o
python3 bench.py --steps 100 --warmup 20 --no-cpu > $o/b4m.log 2>&1
python3 bench.py --lattice 64 --steps 200 --warmup 40 --no-cpu > $o/b1m.log 2>&1
python3 bench.py --lattice 50 --steps 400 --warmup 40 --no-cpu > $o/b500k.log 2>&1
python3 bench.py --lattice 50 --steps 400 --warmup 40 --no-cpu --rccl-loopback > $o/b500k_lb.log 2>&1
python3 bench.py --workload lipid --steps 100 --warmup 20 --no-cpu > $o/blip.log 2>&1
for f in $o/*.log; do echo $f; grep '^{' $f | cut -c1-250; done
