"""Runs one conv variant a few times (for rocprofv3 --pmc).  usage: conv_one.py VARIANT [NBOARDS] [RES]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from c6amd_loader import load_package
pkg = load_package()
v = int(sys.argv[1]); nb = int(sys.argv[2]) if len(sys.argv) > 2 else 8192; res = int(sys.argv[3]) if len(sys.argv) > 3 else 0
ms = pkg.bench_conv(19, 256, nb, 3, variant=v, with_residual=res)
print(f"variant {v}: {ms:.3f} ms, {2*9*256*256*361*nb/ms/1e9:.1f} TFLOP/s")
