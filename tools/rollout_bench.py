"""DefaultPolicy on the GPU: time of Policy::next (num_iter searches) and of bare playouts, per board size.
usage: python tools/rollout_bench.py [games] [num_iter]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from c6amd_loader import load_package  # noqa: E402

pkg = load_package()
G = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
for n in (9, 15, 19):
    dp = pkg.DefaultPolicy(n, G, num_iter=iters, seed=1, max_nodes=8 * iters + 2)
    boards = np.zeros((G, n * n), np.int8)
    boards[:, (n * n) // 2] = -1
    ms, steps, win = dp.bench_playouts(boards, iters=5)
    print(f"N={n} G={G}: playouts {ms:.3f} ms/launch, {steps / G:.1f} stones/playout, {steps / ms / 1e6:.2f} G stones/s, "
          f"{G / ms / 1e3:.2f} M playouts/s, black {np.mean(win == -1):.3f} white {np.mean(win == 1):.3f} draw {np.mean(win == 0):.3f}")
    dp.next(boards)
    t = time.time()
    reps = 4
    for _ in range(reps):
        dp.next(boards)
    dt = (time.time() - t) / reps
    print(f"        next(num_iter={iters}): {dt * 1e3:.1f} ms for {G} policies = {G * iters / dt / 1e6:.2f} M searches/s, {G / dt / 1e3:.1f} k moves/s")
    dp.close()
