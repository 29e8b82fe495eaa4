#!/bin/bash
# The N > 1 code path of bench.py with REAL engines on a ONE-GPU box: RANKS processes share GPU 0, gloo carries the collectives (RCCL wants
# one device per rank).  Default 4: the GPU box admits at most 6 processes on its card and the torchrun launcher counts as one (a 6-rank run
# was killed by the box's process guard: "7 processes had the GPU open"), so the 8-rank case cannot be rehearsed with engines here; its sharding
# plan and all-gather run over gloo on the CPU in tests/test_dist_gloo.py.
#   tools/share_rehearsal.sh [RANKS=4] [GAMES_PER_RANK=512] [STEPS=3]
cd "$(dirname "$0")/.."
export C6_BENCH_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
R=${1:-4}
exec python -m torch.distributed.run --nnodes=1 --nproc-per-node $R --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus $R --backend gloo --games ${2:-512} --steps ${3:-3} --warmup 1
