"""Development tool: times conv variants / ablations in one process (cdna guide rule 24). Needs a GPU."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from c6amd_loader import load_package
pkg = load_package()
L = pkg.lib()
L.c6_bench_conv.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_double)]
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
cases = [(6, 0)]
flops = 2 * 9 * 256 * 256 * 361 * nb
for rnd in range(2):
    for variant, abl in cases:
        for res in (0, 1):
            ms = C.c_double()
            rc = L.c6_bench_conv(19, 256, nb, 3, variant, abl, res, C.byref(ms))
            assert rc == 0, L.c6_last_error()
            print(f"round {rnd} variant {variant} ablate {abl} res {res}: {ms.value:8.3f} ms  {flops / ms.value / 1e9:7.1f} TFLOP/s")
