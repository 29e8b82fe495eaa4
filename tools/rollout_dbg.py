import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from c6amd_loader import load_package
pkg = load_package()
n = 19
for G in (6, 64, 128, 4096):
    dp = pkg.DefaultPolicy(n, G, seed=2)
    boards = np.zeros((G, n * n), np.int8)
    boards[:, 180] = -1
    pos = dp.next(boards)
    print(G, pos[:8], len(np.unique(pos)), [dp.info(g) for g in (0, 1, min(5, G - 1), G - 1)])
    dp.close()
