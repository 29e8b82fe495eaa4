#!/bin/bash
# rocprofv3 kernel trace of the training-step bench (run on the GPU box): per-kernel totals of tools/train_bench.py -> gpurun_out/prof_train/summary.txt
out=gpurun_out/prof_train; mkdir -p $out; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 tools/train_bench.py --steps 3 $* > $out/kt.out 2> $out/kt.err
find $out -name "*_kernel_trace.csv" -delete; find $out -name "*_agent_info.csv" -delete
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$out/kt/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)): rows.append((float(r["TotalDurationNs"]) / 1e6, int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"].split("(")[0]))
rows.sort(reverse=True)
with open("$out/summary.txt", "w") as o:
    for t, c, a, n in rows[:40]: o.write("%9.2f ms total  %6d calls  %9.1f us avg  %s\n" % (t, c, a, n))
print(open("$out/summary.txt").read())
PY
