#!/bin/bash
# exercise the N>1 exchange code (RCCL all-reduce on the library's device vector) with a 1-rank group
cd $GRAFT_REPO_ROOT
KTA_BENCH_FORCE_COLLECTIVES=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 1 --no-cpu-baseline --no-alive --records-per-gpu 134217728
