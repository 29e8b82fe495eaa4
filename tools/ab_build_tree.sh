#!/bin/bash
# A/B library with tree_kernels.hip compiled under extra flags: tools/ab_build_tree.sh NAME -DFLAG...  -> tools/ab/libNAME.so
n=$1; shift
cd "$(dirname "$0")/.." && mkdir -p tools/ab /tmp/ab_$n
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value -Wno-unused-result "$@" -x hip -c alphazero-connect6_amd/csrc/tree_kernels.hip -o /tmp/ab_$n/tree.o || exit 1
objs=$(ls alphazero-connect6_amd/_build/*.o | grep -v tree_kernels)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab/lib$n.so /tmp/ab_$n/tree.o $objs
