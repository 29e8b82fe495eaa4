#!/bin/bash
# The N > 1 code path of bench.py with REAL engines on a ONE-GPU box: RANKS processes (default 6: the GPU box admits at most 6 processes
# on its card, so the 8-rank case cannot be rehearsed with engines here; its sharding plan and all-gather run over gloo on the CPU in
# tests/test_dist_gloo.py) share GPU 0, gloo carries the collectives (RCCL wants one device per rank).
#   tools/share_rehearsal.sh [RANKS=6] [GAMES_PER_RANK=512] [STEPS=3]
cd "$(dirname "$0")/.."
export C6_BENCH_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
R=${1:-6}
exec python -m torch.distributed.run --nnodes=1 --nproc-per-node $R --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus $R --backend gloo --games ${2:-512} --steps ${3:-3} --warmup 1
