#!/bin/bash
# An A/B library from a PATCHED COPY of one source (the tree stays untouched, so bench.kernel_source_hash() does not change):
#   tools/ab_patch.sh NAME FILE.hip tools/ab_patches/SCRIPT.sed [-DFLAG...]   -> tools/ab/libNAME.so   (run with C6_LIB_PATH=$PWD/tools/ab/libNAME.so)
# The ablations behind profiles/r03_tree_ablation.txt: tree_no_noise.sed, tree_no_last_writer.sed, tree_approx_div_sqrt.sed (all on tree_kernels.hip;
# they change the trees or the last bit of the scores: cost measurements only).
n=$1; f=$2; sedf=$3; shift 3
cd "$(dirname "$0")/.." && mkdir -p tools/ab /tmp/ab_$n
sed -f $sedf alphazero-connect6_amd/csrc/$f > /tmp/ab_$n/$f
cmp -s /tmp/ab_$n/$f alphazero-connect6_amd/csrc/$f && { echo "the patch changed nothing"; exit 1; }
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value -Wno-unused-result -I alphazero-connect6_amd/csrc "$@" -x hip -c /tmp/ab_$n/$f -o /tmp/ab_$n/x.o || exit 1
objs=$(ls alphazero-connect6_amd/_build/*.o | grep -v "/$f.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab/lib$n.so /tmp/ab_$n/x.o $objs && echo built tools/ab/lib$n.so
