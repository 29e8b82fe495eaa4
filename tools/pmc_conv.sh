#!/bin/bash
# rocprofv3 SQ counters of one conv variant's microbench (run on the GPU box): tools/pmc_conv.sh VARIANT [RES]
v=$1; r=${2:-0}; out=gpurun_out/pmc_conv_v${v}_r$r; mkdir -p $out; export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $out -- python3 tools/conv_one.py $v 8192 $r > $out/out.txt 2> $out/err.txt
python3 - <<PY
import csv, glob, collections
f = glob.glob("$out/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur[k][r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
for k, c in agg.items():
    if "conv3x3" not in k: continue
    a = {n: sum(v) / len(v) for n, v in c.items()}
    ns = sum(dur[k].values()) / len(dur[k]); cyc = a["GRBM_GUI_ACTIVE"] / 8
    print("v$v res=$r", k[:40], "ms=%.3f clock=%.3f GHz mfma_busy=%.3f wait_any=%.3f wait_inst=%.3f lds_conf=%.3f lds_active/cyc/CU=%.3f valu/wave=%d" % (
        ns / 1e6, cyc / ns, a["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024), a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"], a["SQ_WAIT_INST_ANY"] / a["SQ_WAVE_CYCLES"],
        a["SQ_LDS_BANK_CONFLICT"] / max(a["SQ_LDS_IDX_ACTIVE"], 1), a["SQ_LDS_IDX_ACTIVE"] / (cyc * 256), a["SQ_INSTS_VALU"] / 2048))
PY
