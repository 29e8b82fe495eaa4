#!/bin/bash
# SQ counters of the trainer's kernels (run on the GPU box): tools/pmc_train.sh
out=gpurun_out/pmc_train; mkdir -p $out; export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $out -- python3 tools/train_bench.py --steps 1 > $out/out.txt 2> $out/err.txt
python3 - <<PY
import csv, glob, collections
f = max(glob.glob("$out/**/*counter_collection.csv", recursive=True), key=lambda p: __import__("os").path.getmtime(p))
agg = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"])); dur[k][r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
rows = []
for k, c in agg.items():
    a = {n: sum(v) / len(v) for n, v in c.items()}; ns = sum(dur[k].values()) / len(dur[k]); cyc = a["GRBM_GUI_ACTIVE"] / 8
    rows.append((ns * len(dur[k]), k[:46], len(dur[k]), ns / 1e3, a["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024), cyc / ns, a["SQ_LDS_BANK_CONFLICT"] / max(a["SQ_LDS_IDX_ACTIVE"], 1), a["SQ_WAIT_ANY"] / max(a["SQ_WAVE_CYCLES"], 1)))
for tot, k, n, us, busy, clk, conf, wait in sorted(rows, reverse=True)[:8]:
    print(f"{k:46s} n={n:4d} avg_us={us:9.1f} mfma_busy={busy:.3f} clock={clk:.2f} lds_conflict={conf:.3f} wait={wait:.2f}")
PY
