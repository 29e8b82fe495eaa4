import sys, os
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from c6amd_loader import load_package
pkg = load_package()
import __graft_entry__ as ge
for n, C, blocks in [(19, 256, 4), (15, 256, 10), (9, 64, 3)]:
    rng = np.random.default_rng(n)
    blob = ge.make_random_blob(n, C, blocks, rng)
    boards = rng.integers(-1, 2, size=(64, n * n)).astype(np.int8); player = rng.choice(np.array([-1, 1], np.int8), size=64)
    p1, v1 = pkg.net_forward(n, C, blocks, blob, boards, player, precision=1)
    p3, v3 = pkg.net_forward(n, C, blocks, blob, boards, player, precision=3)
    p4, v4 = pkg.net_forward(n, C, blocks, blob, boards, player, precision=4)
    print(n, C, blocks, "bf16x3 vs fp32: pol %.2e val %.2e | bf16 vs fp32: pol %.2e val %.2e (pol max %.3f)" % (np.abs(p3-p1).max(), np.abs(v3-v1).max(), np.abs(p4-p1).max(), np.abs(v4-v1).max(), p1.max()))
