#!/bin/bash
# same-box comparison of several A/B libraries on one bench workload: tools/ab_many.sh WORKLOAD "NAME1 NAME2 ..." [bench args]   ("tree" = the in-tree library)
w=$1; names=$2; shift 2
for rep in 1 2; do
  for n in $names; do
    lib=$PWD/tools/ab/lib$n.so; [ "$n" = tree ] && lib=""
    echo -n "$n: "
    C6_LIB_PATH=$lib python3 bench.py --workload $w --single-precision --no-cpu-baseline --no-probe "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'] and round(d['value'],4), round(d['ms_per_step'],4), round(d['engine']['select_steps_per_sim'],4))"
  done
done
