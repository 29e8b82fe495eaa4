import sys, os, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
from c6amd_loader import load_package
import __graft_entry__ as ge
pkg = load_package()
for (n, C, blocks) in [(19, 256, 3), (9, 64, 2), (15, 128, 2)]:
    blob = ge.make_random_blob(n, C, blocks, np.random.default_rng(1))
    boards = np.random.default_rng(2).integers(-1, 2, size=(5, n * n)).astype(np.int8)
    ref = pkg.net_forward(n, C, blocks, blob, boards, 1, precision=1)
    got = pkg.net_forward(n, C, blocks, blob, boards, 1, precision=0)   # env C6_CONV_VARIANT decides
    print(n, C, 'max err policy', np.abs(ref[0]-got[0]).max(), 'value', np.abs(ref[1]-got[1]).max())
