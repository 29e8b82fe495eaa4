"""K1 board kernel against the HBM roofline: N*N + 2 algorithmic bytes per board-step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from c6amd_loader import load_package
pkg = load_package()
for n, G in ((9, 32768), (9, 4 << 20), (19, 4096), (19, 1 << 20), (19, 4 << 20)):
    ms = pkg.bench_board(n, G, 30)
    b = (n * n + 2) * G
    print(f"N={n} G={G}: {ms*1e3:9.1f} us/launch  {b/ms/1e6:8.1f} GB/s algorithmic ({b/ms/1e6/8000*100:.1f}% of 8 TB/s)")
