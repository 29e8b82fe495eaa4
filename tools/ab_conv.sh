#!/bin/bash
# same-box A/B of the conv microbench: tools/ab/lib<NAME>.so (tools/ab_build.sh) vs the in-tree library
# usage: [V=variant] tools/ab_conv.sh NAME...
V=${V:-6}
for rep in 1 2; do for r in 0 1; do
  for n in "$@"; do echo -n "$n res=$r: "; C6_LIB_PATH=$PWD/tools/ab/lib$n.so python3 tools/conv_one.py $V 8192 $r; done
  echo -n "tree res=$r: "; python3 tools/conv_one.py $V 8192 $r
done; done
