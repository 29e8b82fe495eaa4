#!/bin/bash
# tools/group_bench.sh [TAG] -- run ON THE GPU BOX (through gpurun): the multi-device host BEHIND the C ABI (csrc/group.cpp) driven by a compiled C++
# program with no Python (tools/cpp/group_selfplay.cpp), at config 2's per-GPU workload: 4096 concurrent 19x19 games, 400 simulations per move, resident
# stem + 3 blocks net, steady state (recycle + stagger), training tuples pooled every 10 rounds with the exchange overlapped.
#   members {0}    : one member, the pooling on REAL RCCL calls (ncclCommInitAll + ncclAllGather on the side stream)
#   members {0,0}  : two members of 2048 games sharing the GPU (two host threads, two engines; host-staged pooling -- RCCL admits one rank per device)
#   all            : every visible device (on an 8-GPU node: 8 x 4096 games over xGMI)
tag=${1:-r06}
out=gpurun_out/group_$tag
mkdir -p $out
lib=$(pwd)/alphazero-connect6_amd
g++ -O2 -std=c++17 -Iinclude tools/cpp/group_selfplay.cpp -L$lib -lconnect6_amd -Wl,-rpath,$lib -o $out/group_selfplay || exit 1
python3 -c "
import numpy as np, __graft_entry__ as ge
ge.make_random_blob(19, 256, 4, np.random.default_rng(0)).astype(np.float32).tofile('$out/blob19.bin')" || exit 1
T="timeout -k 10 300"
ndev=$(python3 -c "import torch; print(torch.cuda.device_count())")
{
  echo "== members {0}: 4096 games, RCCL pooling";          $T $out/group_selfplay 0   19 4096 400 2024 net:$out/blob19.bin:256:4 60 10
  echo "== members {0,0}: 2 x 2048 games, host-staged";     $T $out/group_selfplay 0,0 19 4096 400 2024 net:$out/blob19.bin:256:4 60 10
  if [ "$ndev" -gt 1 ]; then
    echo "== all $ndev devices: $ndev x 4096 games, RCCL";  $T $out/group_selfplay all 19 $((4096 * ndev)) 400 2024 net:$out/blob19.bin:256:4 60 10
  fi
} 2>&1 | tee $out/group_selfplay.txt
rm -f $out/blob19.bin $out/group_selfplay
