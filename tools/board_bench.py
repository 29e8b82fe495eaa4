"""K1 board kernel against the HBM roofline: N*N + 2 algorithmic bytes per board-step; the search-only launch
(is_game_end of every game, no stone) reads N*N and writes 1 byte per board."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from c6amd_loader import load_package
pkg = load_package()
for n, G in ((9, 32768), (9, 4 << 20), (9, 16 << 20), (15, 4 << 20), (19, 4096), (19, 1 << 20), (19, 4 << 20)):
    for scan in (False, True):
        ms = pkg.bench_board(n, G, 30, scan_only=scan)
        b = (n * n + (1 if scan else 2)) * G
        print(f"N={n} G={G} {'search-only' if scan else 'board-step '}: {ms*1e3:9.1f} us/launch  {b/ms/1e6:8.1f} GB/s algorithmic "
              f"({b/ms/1e6/8000*100:.1f}% of 8 TB/s)", flush=True)
