"""Condenses a gpurun_out/prof_rNN directory (rocprofv3 --kernel-trace --stats and two --pmc passes of
`python3 bench.py ...`) into the small files committed under profiles/.

    python profiles/summarize.py gpurun_out/prof_r1 r01
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(here))
import bench  # noqa: E402  (kernel_source_hash: bench.py replays this summary only while the kernel sources are the ones profiled)
stats = max(glob.glob(os.path.join(src, "kt", "**", "*_kernel_stats.csv"), recursive=True), key=os.path.getmtime)
shutil.copy(stats, os.path.join(here, f"{tag}_kernel_stats.csv"))
out = {"command": "rocprofv3 --pmc <COUNTER> --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-probe",
       "src_hash": bench.kernel_source_hash(bench.CONV_SOURCES), "src_hash_of": list(bench.CONV_SOURCES),
       "unit_note": "FETCH_SIZE / WRITE_SIZE are KiB per dispatch; on gfx950 FETCH_SIZE counts 128-B requests at 64 B, so "
                    "hbm_read_bytes = 2 * FETCH_SIZE * 1024 (MI355X_MICROARCH.md, HBM); WRITE_SIZE * 1024 is exact", "kernels": {}}
for name, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    f = max(glob.glob(os.path.join(src, name, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == ctr:
            agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if k.startswith("c6_") or "c6_" in k:
            out["kernels"].setdefault(k, {})[ctr + "_avg_KiB"] = sum(v) / len(v)
            out["kernels"][k][ctr + "_dispatches"] = len(v)
for k, d in out["kernels"].items():
    rd = 2.0 * d.get("FETCH_SIZE_avg_KiB", 0.0) * 1024
    wr = d.get("WRITE_SIZE_avg_KiB", 0.0) * 1024
    d["hbm_bytes_per_launch"] = rd + wr
# optional third pass: SQ / GRBM counters (matrix-core utilisation, effective clock, LDS conflicts)
sq = glob.glob(os.path.join(src, "pmc_sq", "**", "*counter_collection.csv"), recursive=True)
if sq:
    f = max(sq, key=os.path.getmtime)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "c6_" not in k:
            continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r.get("Start_Timestamp") and r.get("End_Timestamp"):
            dur[k][r.get("Dispatch_Id", len(dur[k]))] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    out["sq_note"] = ("SQ_* are sums over all SIMDs of the device; GRBM_GUI_ACTIVE is the sum over the 8 XCDs, so device cycles = "
                      "GRBM_GUI_ACTIVE / 8 (MI355X_MICROARCH.md, DVFS give-back). mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / "
                      "(device cycles * 1024 SIMDs); effective_clock_ghz = device cycles / profiled duration")
    for k, ctrs in agg.items():
        d = out["kernels"].setdefault(k, {})
        for c, v in ctrs.items():
            d[c + "_avg"] = sum(v) / len(v)
        cyc = d.get("GRBM_GUI_ACTIVE_avg", 0.0) / 8.0
        if cyc > 0:
            d["mfma_util"] = d.get("SQ_VALU_MFMA_BUSY_CYCLES_avg", 0.0) / (cyc * 1024.0)
            if dur[k]:
                ns = sum(dur[k].values()) / len(dur[k])
                d["profiled_avg_ns"] = ns
                d["effective_clock_ghz"] = cyc / ns
        if d.get("SQ_LDS_IDX_ACTIVE_avg"):
            d["lds_bank_conflict_frac"] = d.get("SQ_LDS_BANK_CONFLICT_avg", 0.0) / d["SQ_LDS_IDX_ACTIVE_avg"]
for f in glob.glob(os.path.join(src, "bench_*.json")):
    shutil.copy(f, os.path.join(here, f"{tag}_{os.path.basename(f)}"))
json.dump(out, open(os.path.join(here, f"{tag}_pmc_summary.json"), "w"), indent=1)
print(json.dumps(out["kernels"], indent=1)[:3000])
