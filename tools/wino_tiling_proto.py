"""numpy prototype of mixed Winograd tilings of an N x N board (one channel pair): every tile row / column has its own F(m,3) -- the engine's
19 = 4+4+4+4+3 (841 positions per board; 846 in conv_wino4.hip, whose corner tile is F(3) x F(4)) and the candidates of DESIGN 9 (5+5+5+4 and 6+6+4+3: 729).  Checks each tiling
against the direct convolution in float64 and prints positions per board and the transform matrices.
    python tools/wino_tiling_proto.py [N=19]"""
import sys
import numpy as np
from wino_err import cook_toom

POINTS = {3: [0, 1, -1, 2], 4: [0, 1, -1, 2, -2], 5: [0, 1, -1, 2, -2, "1/2"], 6: [0, 1, -1, 2, -2, "1/2", "-1/2"]}
MATS = {m: cook_toom(p, m) for m, p in POINTS.items()}      # m -> (AT, G, BT)


def run(N, split, rng):
    assert sum(split) == N
    x = rng.standard_normal((N, N)); g = rng.standard_normal((3, 3))
    xp = np.zeros((N + 2, N + 2)); xp[1:N + 1, 1:N + 1] = x
    ref = np.array([[sum(g[a, b] * xp[r + a, c + b] for a in range(3) for b in range(3)) for c in range(N)] for r in range(N)])
    out = np.zeros((N, N)); npos = 0
    r0 = 0
    for mv in split:
        c0 = 0
        for mh in split:
            ATv, Gv, BTv = MATS[mv]; ATh, Gh, BTh = MATS[mh]
            d = xp[r0:r0 + mv + 2, c0:c0 + mh + 2]
            M = (Gv @ g @ Gh.T) * (BTv @ d @ BTh.T)
            npos += M.size
            out[r0:r0 + mv, c0:c0 + mh] = ATv @ M @ ATh.T
            c0 += mh
        r0 += mv
    return npos, np.abs(out - ref).max()


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 19
    rng = np.random.default_rng(0)
    splits = {19: ([4, 4, 4, 4, 3], [5, 5, 5, 4], [6, 6, 4, 3], [4, 4, 4, 4, 4]), 15: ([4, 4, 4, 3], [5, 5, 5], [6, 6, 3])}.get(N, ([4] * (N // 4) + ([N % 4] if N % 4 else []),))
    for sp in splits:
        if sum(sp) != N:      # the all-F(4) tiling computes a row and a column beyond the board
            print(sp, "positions per board", sum(m + 2 for m in sp) ** 2, "(padded)")
            continue
        npos, err = run(N, sp, rng)
        print(sp, "positions per board", npos, "max error vs direct (float64)", err)
    np.set_printoptions(linewidth=200, precision=6, suppress=True)
    for m in (5,):
        AT, G, BT = MATS[m]
        print(f"F({m},3): A^T =\n{AT}\nG =\n{G}\nB^T =\n{BT}")
