#!/bin/bash
# HBM traffic of the K1 board kernel in its microbench (run on the GPU box): tools/pmc_board.sh N G
# FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950.
n=${1:-19}; g=${2:-4194304}; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  out=gpurun_out/pmc_board_${n}_$c; mkdir -p $out
  rocprofv3 --pmc $c --output-format csv -d $out -- python3 tools/board_one.py $n $g > $out/out.txt 2> $out/err.txt
  python3 - <<PY
import csv, glob
f = glob.glob("$out/**/*counter_collection.csv", recursive=True)[0]
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "c6_board_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "$c"]
k = 2 if "$c" == "FETCH_SIZE" else 1
print("N=$n G=$g $c avg KiB %.0f -> %.1f bytes per board%s (algorithmic %d)" % (sum(v)/len(v), sum(v)/len(v)*1024*k/$g, " (x2 applied)" if k == 2 else "", $n*$n+2))
PY
done
