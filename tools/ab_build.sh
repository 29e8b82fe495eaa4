#!/bin/bash
# A/B library with ONE source compiled under extra flags: tools/ab_build.sh NAME FILE.hip -DFLAG...  -> tools/ab/libNAME.so
# (the other objects come from the in-tree build; run with C6_LIB_PATH=$PWD/tools/ab/libNAME.so)
n=$1; f=$2; shift 2
cd "$(dirname "$0")/.." && mkdir -p tools/ab /tmp/ab_$n
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value -Wno-unused-result "$@" -x hip -c alphazero-connect6_amd/csrc/$f -o /tmp/ab_$n/x.o || exit 1
objs=$(ls alphazero-connect6_amd/_build/*.o | grep -v "/$f.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab/lib$n.so /tmp/ab_$n/x.o $objs
