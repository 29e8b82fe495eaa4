#!/bin/bash
# rocprofv3 SQ counters of the tree kernels on config 5 (run on the GPU box); optional C6_LIB_PATH for an A/B library
out=gpurun_out/pmc_tree_${1:-tree}; mkdir -p $out; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $out -- python3 bench.py --workload config5 --steps 10 --warmup 5 --single-precision --no-cpu-baseline --no-probe > $out/bench.json 2> $out/err.txt
tail -2 $out/err.txt
python3 - <<PY
import csv, glob, collections
f = glob.glob("$out/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur[k][r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
for k, c in agg.items():
    if "c6_" not in k: continue
    print(k, "n=%d avg_us=%.1f" % (len(dur[k]), sum(dur[k].values()) / len(dur[k]) / 1e3), {n: int(sum(v) / len(v)) for n, v in c.items()}, "VGPR", "")
PY
