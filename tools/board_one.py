"""one K1 microbench configuration (for rocprofv3): board_one.py N G [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from c6amd_loader import load_package
pkg = load_package()
n, G = int(sys.argv[1]), int(sys.argv[2])
it = int(sys.argv[3]) if len(sys.argv) > 3 else 10
ms = pkg.bench_board(n, G, it)
b = (n * n + 2) * G
print(f"N={n} G={G}: {ms*1e3:9.1f} us/launch  {b/ms/1e6:8.1f} GB/s algorithmic ({b/ms/1e6/8000*100:.1f}% of 8 TB/s)")
