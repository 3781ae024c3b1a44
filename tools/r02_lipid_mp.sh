#!/bin/bash
# bench.py --workload lipid on 2 and 8 ranks sharing the one GPU (host transport) against one rank
cd $GRAFT_REPO_ROOT
export DDCMI_BENCH_SINGLE_DEVICE=1 DDCMI_TRANSPORT=host
one=$(python3 bench.py --no-cpu --workload lipid --reps 4,4,2 --steps 40 --warmup 10 2>&1 | grep '^{')
echo "$one" | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('1 rank :', d['config']['beads_total'], d['check'])"
for n in 2 8; do
  out=$(timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500+n)) bench.py --gpus $n --workload lipid --reps 4,4,2 --steps 40 --warmup 10 2>&1 | grep '^{')
  echo "$out" | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$n ranks:', d['config']['beads_total'], d['check'], d['ms_per_step'])"
done
