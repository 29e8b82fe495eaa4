#!/bin/bash
# tools/ab_build.sh NAME [SRC]: links tools/ab/libNAME.so from the in-tree objects with conv_bf16x3 compiled from SRC
# (default: the working-tree file) -- for same-box A/B runs through C6_LIB_PATH.
set -e
cd "$(dirname "$0")/.."
name=$1; src=${2:-alphazero-connect6_amd/csrc/conv_bf16x3.hip}
mkdir -p tools/ab
cp "$src" alphazero-connect6_amd/csrc/_ab_tmp.hip
(cd alphazero-connect6_amd/csrc && hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -I../../include -Wno-unused-value -x hip -c _ab_tmp.hip -o /tmp/ab_$name.o; rm -f _ab_tmp.hip)
B=alphazero-connect6_amd/_build
hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab/lib$name.so $(ls $B/*.o | grep -v conv_bf16x3) /tmp/ab_$name.o
echo built tools/ab/lib$name.so
