"""K1 board-step ablations (tools/ab_build.sh ... board_kernel.hip -DC6_AB_BOARD_NO_STONE / -DC6_AB_BOARD_NO_STATUS): which part of a
board-step keeps it below the search-only launch.  python tools/board_ab.py  (each library in a child process through C6_LIB_PATH)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = "import sys; sys.path.insert(0, %r)\nfrom c6amd_loader import load_package\npkg = load_package()\nfor n, G in ((19, 1 << 20), (19, 1 << 22), (9, 1 << 24)):\n    a = pkg.bench_board(n, G, 30); b = pkg.bench_board(n, G, 30, scan_only=True)\n    print('  N=%%d G=%%d board-step %%7.1f us (%%4.1f %%%% of 8 TB/s)  search-only %%7.1f us (%%4.1f %%%%)' %% (n, G, a * 1e3, (n*n+2)*G/a/1e6/80, b * 1e3, (n*n+1)*G/b/1e6/80))\n" % ROOT
for name in ("", "nostone", "nostatus", "noboth"):
    lib = os.path.join(ROOT, "tools", "ab", f"lib{name}.so") if name else ""
    print(name or "tree (as built)", flush=True)
    subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, C6_LIB_PATH=lib) if lib else dict(os.environ), check=True)
