#!/bin/bash
# same-box A/B of a bench workload: tools/ab_bench.sh NAME WORKLOAD [bench args]; NAME = tools/ab/lib<NAME>.so vs in-tree
n=$1; w=$2; shift 2
for rep in 1 2; do
  for lib in $PWD/tools/ab/lib$n.so ""; do
    echo -n "${lib:-tree}: "
    C6_LIB_PATH=$lib python3 bench.py --workload $w --single-precision --no-cpu-baseline --no-probe "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],4), round(d['ms_per_step'],4), {k:round(v,3) for k,v in d['phase_ms'].items()})"
  done
done
