#!/bin/bash
# 2-rank rehearsal of bench.py --rehearsal on a one-GPU box (both ranks on cuda:0, collectives over gloo): the sharded flow end to end --
# balanced shards, split ticks, async exchange, re-pack -- its timings mean nothing.   bash tools/rehearse_sharded.sh [workload] [ranks]
cd $GRAFT_REPO_ROOT
python -m torch.distributed.run --nnodes=1 --nproc-per-node ${2:-2} --master-addr 127.0.0.1 --master-port 29611 \
    bench.py --rehearsal --gpus ${2:-2} --workload ${1:-c3} --steps 70 --warmup 10 --windows 2 --min-seconds 0.1 --no-cpu-baseline
