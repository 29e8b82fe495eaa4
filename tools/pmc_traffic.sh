#!/bin/bash
# HBM traffic of one conv variant's microbench (run on the GPU box): tools/pmc_traffic.sh VARIANT RES
v=$1; r=${2:-0}; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  out=gpurun_out/pmc_traffic_v${v}_r${r}_$c; mkdir -p $out
  rocprofv3 --pmc $c --output-format csv -d $out -- python3 tools/conv_one.py $v 8192 $r > $out/out.txt 2> $out/err.txt
  python3 - <<PY
import csv, glob
f = glob.glob("$out/**/*counter_collection.csv", recursive=True)[0]
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "conv3x3" in r["Kernel_Name"] and r["Counter_Name"] == "$c"]
print("v$v res=$r $c avg KiB %.0f -> %.2f GB%s" % (sum(v)/len(v), sum(v)/len(v)*1024/1e9*(2 if "$c"=="FETCH_SIZE" else 1), " (x2 applied)" if "$c"=="FETCH_SIZE" else ""))
PY
done
