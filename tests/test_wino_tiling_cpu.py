"""The Winograd constants of the FP32 evaluator (csrc/c6_wino.h, csrc/conv_wino6.hip, csrc/conv_wino4.hip), written out here as the kernels
use them, against the direct 3x3 convolution in float64 (model.hpp:61-78: conv3x3, padding 1) -- one channel pair, every tile class of the
19x19 tiling of conv_wino6.hip (columns 6 + 6 + 4 + 3, rows 4 + 4 + 4 + 4 + 3) and of the
F(4,3) / F(3,3) tiling of conv_wino4.hip.  Host arithmetic only: it pins the matrices and the position counts (788 / 846 per channel pair and
19x19 board, 534 at 15x15), the kernels themselves are held to 1e-5 against the oracle in tests/test_gpu_net.py."""
import numpy as np

# F(4,3), points 0, +-1, +-2, inf (Lavin & Gray)
BT4 = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], float)
G4 = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]])
AT4 = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], float)
# F(3,3), points 0, +-1, 2, inf
BT3 = np.array([[2, -1, -2, 1, 0], [0, -2, -1, 1, 0], [0, 2, -3, 1, 0], [0, -1, 0, 1, 0], [0, 2, -1, -2, 1]], float)
G3 = np.array([[1 / 2, 0, 0], [-1 / 2, -1 / 2, -1 / 2], [-1 / 6, 1 / 6, -1 / 6], [1 / 6, 1 / 3, 2 / 3], [0, 0, 1]])
AT3 = np.array([[1, 1, 1, 1, 0], [0, 1, -1, 2, 0], [0, 1, 1, 4, 1]], float)
# F(6,3), points 0, +-1, +-2, +-1/2, inf
BT6 = np.array([[1, 0, -21 / 4, 0, 21 / 4, 0, -1, 0], [0, 1, 1, -17 / 4, -17 / 4, 1, 1, 0], [0, -1, 1, 17 / 4, -17 / 4, -1, 1, 0],
                [0, 1 / 2, 1 / 4, -5 / 2, -5 / 4, 2, 1, 0], [0, -1 / 2, 1 / 4, 5 / 2, -5 / 4, -2, 1, 0], [0, 2, 4, -5 / 2, -5, 1 / 2, 1, 0],
                [0, -2, 4, 5 / 2, -5, -1 / 2, 1, 0], [0, -1, 0, 21 / 4, 0, -21 / 4, 0, 1]])
G6 = np.array([[1, 0, 0], [-2 / 9, -2 / 9, -2 / 9], [-2 / 9, 2 / 9, -2 / 9], [1 / 90, 1 / 45, 2 / 45], [1 / 90, -1 / 45, 2 / 45],
               [32 / 45, 16 / 45, 8 / 45], [32 / 45, -16 / 45, 8 / 45], [0, 0, 1]])
AT6 = np.array([[1, 1, 1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 1 / 2, -1 / 2, 0], [0, 1, 1, 4, 4, 1 / 4, 1 / 4, 0], [0, 1, -1, 8, -8, 1 / 8, -1 / 8, 0],
                [0, 1, 1, 16, 16, 1 / 16, 1 / 16, 0], [0, 1, -1, 32, -32, 1 / 32, -1 / 32, 1]])
F = {3: (AT3, G3, BT3), 4: (AT4, G4, BT4), 6: (AT6, G6, BT6)}


def direct(x, g):
    n = x.shape[0]
    xp = np.zeros((n + 2, n + 2))
    xp[1:-1, 1:-1] = x
    return np.array([[(g * xp[r:r + 3, c:c + 3]).sum() for c in range(n)] for r in range(n)]), xp


def tiled(xp, g, n, rows, cols, forms):
    """rows / cols: (first output, outputs covered, F(m) used) per tile row / column; forms(ty, tx, mv, mh) -> the (vertical, horizontal)
    transforms of that tile (the classes that deviate from their row's / column's)"""
    out = np.zeros((n, n))
    npos = 0
    for ty, (r0, hr, mv0) in enumerate(rows):
        for tx, (c0, wc, mh) in enumerate(cols):
            mv, mh = forms(ty, tx, mv0, mh)
            ATv, Gv, BTv = F[mv]
            ATh, Gh, BTh = F[mh]
            d = np.zeros((mv + 2, mh + 2))
            rr, cc = min(mv + 2, n + 2 - r0), min(mh + 2, n + 2 - c0)      # patches may reach past the padded board: zeros
            d[:rr, :cc] = xp[r0:r0 + rr, c0:c0 + cc]
            m = (Gv @ g @ Gh.T) * (BTv @ d @ BTh.T)
            npos += m.size
            y = ATv @ m @ ATh.T
            out[r0:r0 + hr, c0:c0 + wc] = y[:hr, :wc]                      # outputs beyond the tile's share of the board are dropped
    return out, npos


def test_wino6_tiling_of_the_19_board():
    rng = np.random.default_rng(1)
    n = 19
    x, g = rng.standard_normal((n, n)), rng.standard_normal((3, 3))
    ref, xp = direct(x, g)
    cols = [(0, 6, 6), (6, 6, 6), (12, 4, 4), (16, 3, 3)]
    rows = [(0, 4, 4), (4, 4, 4), (8, 4, 4), (12, 4, 4), (16, 3, 3)]
    # r4: the six-wide tiles of the last tile row are class D' (F(3,3) down, F(6,3) across: 5 x 8 positions); the corner tile is class B'
    # (F(3,3) down, F(4,3) across, three of its four columns used)
    out, npos = tiled(xp, g, n, rows, cols, lambda ty, tx, mv, mh: (mv, 4 if (ty == 4 and tx == 3) else mh))
    assert npos == 788 == 8 * 48 + 2 * 40 + 4 * 36 + 4 * 30 + 2 * 30
    assert np.abs(out - ref).max() < 1e-11
    # round 3's tiling (class D kept F(4,3) down the board in the last tile row: its fourth output row lies beyond the board)
    out, npos = tiled(xp, g, n, rows, cols, lambda ty, tx, mv, mh: (4 if tx < 2 else mv, 4 if (ty == 4 and tx == 3) else mh))
    assert npos == 804 and np.abs(out - ref).max() < 1e-11


def test_wino6_tiling_of_the_15_board():
    """r6: conv_wino6.hip's tiling of the reference's 15x15 board (W6G<15>): columns 6 + 6 + 3, rows 4 + 4 + 4 + 3; classes D (F(4) down x F(6) across, 6 tiles),
    D' (F(3) x F(6), 2), B (F(4) x F(3), 3) and the corner as B' (F(3) down, F(4) across, three of its four columns used: the class it is at 19x19).
    488 positions per channel pair and board -- conv_wino4.hip's F(4x4) tiling of the same board has 534."""
    rng = np.random.default_rng(4)
    n = 15
    x, g = rng.standard_normal((n, n)), rng.standard_normal((3, 3))
    ref, xp = direct(x, g)
    cols = [(0, 6, 6), (6, 6, 6), (12, 3, 3)]
    rows = [(0, 4, 4), (4, 4, 4), (8, 4, 4), (12, 3, 3)]
    out, npos = tiled(xp, g, n, rows, cols, lambda ty, tx, mv, mh: (mv, 4 if (ty == 3 and tx == 2) else mh))
    assert npos == 488 == 6 * 48 + 2 * 40 + 3 * 30 + 30
    assert np.abs(out - ref).max() < 1e-11


def test_wino4_tiling_of_the_19_and_15_boards():
    rng = np.random.default_rng(2)
    for n, want in ((19, 846), (15, 534)):      # 16 x 36 + 9 x 30 and 9 x 36 + 7 x 30
        x, g = rng.standard_normal((n, n)), rng.standard_normal((3, 3))
        ref, xp = direct(x, g)
        k = n // 4
        span = [(4 * i, 4, 4) for i in range(k)] + [(4 * k, 3, 3)]
        # the corner tile is class B' (F(3,3) down, F(4,3) across, three of its four columns used)
        last = len(span) - 1
        out, npos = tiled(xp, g, n, span, span, lambda ty, tx, mv, mh: (mv, 4 if (ty == last and tx == last) else mh))
        assert npos == want
        assert np.abs(out - ref).max() < 1e-11


def test_z_rows_are_half_of_the_output_transform():
    """the GEMM kernels accumulate Z_i[b] = sum_j A4^T[b][j] M[i][j] along the 6-long direction of a row of positions; the boundary kernels
    apply the other direction's A^T to the rows: A_v^T (M A_h) = A_v^T Z for every class"""
    rng = np.random.default_rng(3)
    for mv, mh in ((4, 4), (6, 4), (3, 4)):
        m = rng.standard_normal((mv + 2, 6))
        z = m @ F[4][0].T                                   # [rows of positions][b = 4]
        assert z.shape == (mv + 2, 4)
        assert np.abs(F[mv][0] @ z - F[mv][0] @ m @ F[4][0].T).max() < 1e-12
    # class D' (r4): rows of FIVE positions, Z_i[b] = sum_j A3^T[b][j] M[i][j], three row sums per row of positions; the GEMM kernel keeps the third in
    # the registers (and in Z's slot b = 3) of F(4)'s fourth: built from the points (0, 1, -1, 2, inf) exactly as W6Z_UPDATE3 does
    m = rng.standard_normal((8, 5))
    z0 = m[:, 0] + m[:, 1] + m[:, 2] + m[:, 3]
    z1 = m[:, 1] - m[:, 2] + 2 * m[:, 3]
    z3 = m[:, 1] + m[:, 2] + 4 * m[:, 3] + m[:, 4]
    assert np.abs(np.stack([z0, z1, z3], 1) - m @ F[3][0].T).max() < 1e-12


def test_candidate_tilings_position_counts():
    """DESIGN 9, lever L2 (not built): F(6,3) both ways on the 19x19 board -- 19 = 6 + 6 + 4 + 3 down as well as across -- is exact and takes 27^2 = 729
    positions per channel pair and board (today 788: -7.5 % matrix work; its f32 error is 1.4 x today's, tools/wino_tile_error.py); the reference's
    own 15x15 board as 6 + 6 + 3 both ways would take 21^2 = 441 (today's F(4,3) tiling: 534), and with F(4,3) kept down the board 23 x 21 = 483."""
    rng = np.random.default_rng(3)
    same = lambda ty, tx, mv, mh: (mv, mh)
    for n, rows, cols, want in ((19, [(0, 6, 6), (6, 6, 6), (12, 4, 4), (16, 3, 3)], [(0, 6, 6), (6, 6, 6), (12, 4, 4), (16, 3, 3)], 729),
                                (15, [(0, 6, 6), (6, 6, 6), (12, 3, 3)], [(0, 6, 6), (6, 6, 6), (12, 3, 3)], 441),
                                (15, [(0, 4, 4), (4, 4, 4), (8, 4, 4), (12, 3, 3)], [(0, 6, 6), (6, 6, 6), (12, 3, 3)], 483)):
        x, g = rng.standard_normal((n, n)), rng.standard_normal((3, 3))
        ref, xp = direct(x, g)
        out, npos = tiled(xp, g, n, rows, cols, same)
        assert npos == want and np.abs(out - ref).max() < 1e-10, (n, npos)
