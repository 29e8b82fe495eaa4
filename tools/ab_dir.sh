#!/bin/bash
# A/B library from a patched COPY of the whole csrc directory (headers included): tools/ab_dir.sh NAME FILE.hip 'shell command run inside the copy' [-DFLAG...]
#   -> tools/ab/libNAME.so (the other objects come from the in-tree build; run with C6_LIB_PATH=$PWD/tools/ab/libNAME.so)
n=$1; f=$2; cmd=$3; shift 3
cd "$(dirname "$0")/.." && mkdir -p tools/ab && rm -rf /tmp/abd_$n && mkdir -p /tmp/abd_$n/alphazero-connect6_amd && cp -r alphazero-connect6_amd/csrc /tmp/abd_$n/alphazero-connect6_amd/csrc && cp -r include /tmp/abd_$n/include
(cd /tmp/abd_$n/alphazero-connect6_amd/csrc && eval "$cmd") || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value -Wno-unused-result "$@" -x hip -c /tmp/abd_$n/alphazero-connect6_amd/csrc/$f -o /tmp/abd_$n/x.o || exit 1
objs=$(ls alphazero-connect6_amd/_build/*.o | grep -v "/$f.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab/lib$n.so /tmp/abd_$n/x.o $objs && echo built tools/ab/lib$n.so
