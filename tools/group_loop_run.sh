#!/bin/bash
# run on the GPU box: the reference trainer's outer loop (AlphaZero/cpp_weighted/main.cpp:138-168) at the reference's own shape -- 15x15, C = 256, stem + 9 blocks,
# mini-batch 2048 -- as a compiled C++ host over a c6_group (tools/cpp/group_loop.cpp), two epochs of 512 games at 8 simulations per move (short games' worth of
# self-play: the point is the loop, not the strength), members {0, 0} (rehearsal on one GPU) and {0} (RCCL pooling)
out=gpurun_out/group_loop; mkdir -p $out
lib=$(pwd)/alphazero-connect6_amd
g++ -O2 -std=c++17 -Iinclude tools/cpp/group_loop.cpp -L$lib -lconnect6_amd -Wl,-rpath,$lib -o $out/group_loop || exit 1
{
  echo "== members {0,0}: 15x15, C=256, stem+9 blocks, 512 games, 8 sims/move, 2 epochs x 8 SGD steps of 2048"
  timeout -k 10 500 $out/group_loop 0,0 15 256 10 512 8 2 8 2048
  echo "== members {0}"
  timeout -k 10 500 $out/group_loop 0 15 256 10 512 8 2 8 2048
} 2>&1 | tee $out/group_loop.txt
rm -f $out/group_loop
