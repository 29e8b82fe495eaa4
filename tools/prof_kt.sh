#!/bin/bash
# rocprofv3 --kernel-trace --stats of a short bench run (run on the GPU box): tools/prof_kt.sh TAG [bench args]   (C6_LIB_PATH: an A/B library)
# C6_NO_VMM=1: under rocprofv3 the pools are plain allocations (a run with hipMemMap'ed pools under the profiler did not come back)
tag=${1:-kt}; shift
export TMPDIR=/tmp C6_NO_VMM=1
out=$PWD/gpurun_out/prof_$tag; mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --single-precision --no-cpu-baseline --no-probe --steps 3 --warmup 1 "$@" > $out/bench.json 2> $out/err.txt
echo "rocprof rc $?"
python3 - <<PY
import csv, glob
f = glob.glob("$out/**/*kernel_stats.csv", recursive=True)
for r in list(csv.DictReader(open(f[0])))[:7] if f else []:
    print("  %-64s calls %3s  avg %8.3f ms" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
