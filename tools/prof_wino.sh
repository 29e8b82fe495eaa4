#!/bin/bash
# rocprofv3 passes over the FP32 convolution microbench (run on the GPU box): tools/prof_wino.sh [BOARDS] [PRECISION]
#   kernel trace -> per-kernel durations; SQ pass -> matrix-core busy, clock, waits, LDS; TCC passes -> HBM bytes
nb=${1:-32768}; prec=${2:-1}; out=gpurun_out/prof_wino_p${prec}_$nb; mkdir -p $out; export TMPDIR=/tmp
B="python3 tools/wino_bench.py $nb 3 $prec"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- $B > $out/kt.out 2> $out/kt.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $out/sq -- $B > $out/sq.out 2> $out/sq.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- $B > $out/fetch.out 2> $out/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- $B > $out/write.out 2> $out/write.err
find $out -name "*_kernel_trace.csv" -delete; find $out -name "*_agent_info.csv" -delete
python3 - <<PY
import csv, glob, collections, json
res = {"boards": $nb, "precision": $prec, "kernels": {}}
for f in glob.glob("$out/kt/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Name"].split("(")[0]
        if "fill_random" in k: continue
        res["kernels"].setdefault(k, {})["avg_ms"] = float(r["AverageNs"]) / 1e6
        res["kernels"][k]["calls"] = int(r["Calls"])
def counters(sub):
    agg = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(dict)
    for f in glob.glob("$out/%s/**/*counter_collection.csv" % sub, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[k][r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return agg, dur
agg, dur = counters("sq")
for k, c in agg.items():
    if k not in res["kernels"]: continue
    a = {n: sum(v) / len(v) for n, v in c.items()}
    ns = sum(dur[k].values()) / len(dur[k]); cyc = a["GRBM_GUI_ACTIVE"] / 8
    res["kernels"][k].update(pmc_ms=ns / 1e6, clock_ghz=cyc / ns, mfma_busy=a["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024),
                             wait_any=a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"], wait_inst=a["SQ_WAIT_INST_ANY"] / a["SQ_WAVE_CYCLES"],
                             lds_conflict=a["SQ_LDS_BANK_CONFLICT"] / max(a["SQ_LDS_IDX_ACTIVE"], 1), lds_active_per_cu_cycle=a["SQ_LDS_IDX_ACTIVE"] / (cyc * 256),
                             valu_per_wave=a["SQ_INSTS_VALU"] / max(a["SQ_WAVE_CYCLES"], 1))
for sub, name, mul in (("fetch", "FETCH_SIZE", 2.0), ("write", "WRITE_SIZE", 1.0)):      # FETCH_SIZE counts half the bytes of wide reads on gfx950 (MI355X_MICROARCH.md); both in KB
    agg, dur = counters(sub)
    for k, c in agg.items():
        if k in res["kernels"] and name in c:
            res["kernels"][k]["hbm_%s_gb" % sub] = mul * sum(c[name]) / len(c[name]) * 1024 / 1e9
s = json.dumps(res, indent=1); open("$out/summary.json", "w").write(s + "\n"); print(s)
PY
