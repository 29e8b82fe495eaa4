"""average launch time of the 3x3 convolution of every C6_PREC_* arithmetic (c6_bench_conv, random data):
    python tools/conv_bench.py [boards=8192] [iters=5]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from c6amd_loader import load_package

pkg = load_package()
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n, C = 19, 256
flop = 2 * 9 * C * C * n * n * nb
out = {}
for name, prec in (("fp32_winograd", pkg.PREC_FP32), ("fp32_direct", pkg.PREC_FP32_DIRECT), ("bf16x3", pkg.PREC_BF16X3), ("f16f8", pkg.PREC_F16F8)):
    for res in (0, 1):
        ms = pkg.bench_conv(n, C, nb, iters=iters, precision=prec, with_residual=res)
        out[f"{name}{'_res' if res else ''}"] = {"ms": round(ms, 3), "algorithmic_tflops": round(flop / ms / 1e9, 1)}
print(json.dumps({"boards": nb, "n": n, "C": C, "conv": out}))
