#!/bin/bash
# HBM bytes of the tree kernels on config 5 (run on the GPU box): FETCH_SIZE and WRITE_SIZE in separate passes (TCC slots)
out=gpurun_out/pmc_tree_traffic_${1:-r02}; mkdir -p $out; export TMPDIR=/tmp
B="python3 bench.py --workload config5 --steps 10 --warmup 5 --single-precision --no-cpu-baseline --no-probe"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- $B > $out/fetch.json 2> $out/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- $B > $out/write.json 2> $out/write.err
python3 - <<PY
import csv, glob, collections
for sub, name in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    agg = collections.defaultdict(list); dur = collections.defaultdict(list)
    for f in glob.glob("$out/%s/**/*counter_collection.csv" % sub, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != name: continue
            k = r["Kernel_Name"].split("(")[0]
            agg[k].append(float(r["Counter_Value"])); dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    for k in agg:
        if "c6_" in k:      # counter unit: KB; FETCH_SIZE x2 only for wide (16 B/lane) streaming reads (MI355X_MICROARCH.md) -- these kernels read 2..16 B per lane: raw value shown
            print("%-10s %-40s n=%d avg_us=%.1f avg_MB=%.1f" % (name, k[:40], len(agg[k]), sum(dur[k]) / len(dur[k]) / 1e3, sum(agg[k]) / len(agg[k]) * 1024 / 1e6))
PY
