#!/bin/bash
# address-translation counters of the tree kernels on config 5 (run on the GPU box): UTCL1 requests / hits / misses per kernel
out=gpurun_out/pmc_tlb; mkdir -p $out; export TMPDIR=/tmp
rocprofv3 --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE --output-format csv -d $out -- python3 bench.py --workload config5 --steps 10 --warmup 5 --single-precision --no-cpu-baseline > $out/bench.json 2> $out/err.txt
tail -2 $out/err.txt
python3 - <<PY
import csv, glob, collections
f = glob.glob("$out/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    if "c6_" in k: print(k, {n: int(sum(v) / len(v)) for n, v in c.items()})
PY
