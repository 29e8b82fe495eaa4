#!/bin/bash
# tools/ab_build_m8.sh NAME SRC: links tools/ab/libNAME.so from the in-tree objects with conv_m8 compiled from SRC -- for same-box
# A/B runs through C6_LIB_PATH (devices of the pool differ by ~5 %, so only same-box comparisons count)
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2
mkdir -p tools/ab
cp "$src" alphazero-connect6_amd/csrc/_ab_tmp.hip
(cd alphazero-connect6_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -I../../include -Wno-unused-value -Wno-unused-result -x hip -c _ab_tmp.hip -o /tmp/ab_$name.o; rm -f _ab_tmp.hip)
B=alphazero-connect6_amd/_build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab/lib$name.so $(ls $B/*.o | grep -v conv_m8) /tmp/ab_$name.o
echo built tools/ab/lib$name.so
