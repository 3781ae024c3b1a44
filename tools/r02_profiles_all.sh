#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/profile_r02.sh 4m --steps 100 --warmup 20
bash tools/profile_r02.sh 1m --lattice 64 --steps 200 --warmup 40
bash tools/profile_r02.sh brick --lattice 50 --steps 400 --warmup 40 --rccl-loopback
bash tools/profile_r02.sh lipid --workload lipid --steps 100 --warmup 20
