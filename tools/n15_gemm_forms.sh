#!/bin/bash
# run on the GPU box: the F(4x4) GEMM kernel (conv_wino4.hip) at 15x15 on the reference trainer's batch (2048 boards = 512 blocks), per work decomposition
export TMPDIR=/tmp
for form in 0 1; do
  out=gpurun_out/prof_n15gemm_$form; mkdir -p $out
  C6_W4_SPLIT=$form timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 -c "
import sys; sys.path.insert(0, '.')
from c6amd_loader import load_package
pkg = load_package()
print('%.3f ms per layer' % pkg.bench_conv(15, 256, 2048, iters=20, precision=pkg.PREC_FP32, with_residual=0))" 2> $out/err.txt
  python3 - <<PY
import csv, glob
f = glob.glob("$out/**/*kernel_stats.csv", recursive=True)
for r in list(csv.DictReader(open(f[0])))[:5] if f else []:
    print("  C6_W4_SPLIT=$form  %-60s calls %3s  avg %8.3f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
