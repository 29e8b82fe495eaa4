#!/bin/bash
# rocprofv3 --kernel-trace --stats of a short bench run (run on the GPU box): tools/prof_kt.sh TAG [bench args]
# C6_NO_VMM=1: under rocprofv3 the pools are plain allocations (a run with hipMemMap'ed pools under the profiler did not come back)
tag=${1:-kt}; shift
export TMPDIR=/tmp C6_NO_VMM=1
out=$PWD/gpurun_out/prof_$tag; mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --single-precision --no-cpu-baseline --no-probe --steps 3 --warmup 1 "$@" > $out/bench.json 2> $out/err.txt
echo "rocprof rc $?"
f=$(find $out -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cut -d, -f1-4,7 $f | head -16
