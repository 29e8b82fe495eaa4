"""the FP32 Winograd convolution alone at the benchmark's batch (for rocprofv3 --kernel-trace / --pmc):
    python tools/wino_bench.py [boards=32768] [iters=4] [precision=1]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from c6amd_loader import load_package

pkg = load_package()
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 4
prec = int(sys.argv[3]) if len(sys.argv) > 3 else pkg.PREC_FP32
for res in (0, 1):
    ms = pkg.bench_conv(19, 256, nb, iters=iters, precision=prec, with_residual=res)
    print(f"precision {prec} boards {nb} residual {res}: {ms:.3f} ms = {2 * 9 * 256 * 256 * 361 * nb / ms / 1e9:.1f} TFLOP/s algorithmic")
