"""kernel-trace helper: the FP32 Winograd layer a few times (use under rocprofv3 --kernel-trace --stats, or read the total)
    python tools/wino_gemm_only.py [boards=32768] [iters=3]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from c6amd_loader import load_package
pkg = load_package()
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
print("%.3f ms per layer" % pkg.bench_conv(19, 256, nb, iters=iters, precision=pkg.PREC_FP32, with_residual=0))
