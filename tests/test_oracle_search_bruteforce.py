"""The oracle's winner scan against an INDEPENDENT restatement of the reference's rule (search_winner/mod.rs:201-246), on random boards.

The reference scans the board row-major and, at every stone, extends the run that ends there in four directions -- from the left neighbour,
from above, from the upper left, from the upper right -- and returns that stone's colour the moment a run reaches six (overlines count).
The oracle restates the reference's two-row dynamic programme; this file counts every run directly, cell by cell, with no table at all,
so a slip in either is caught by the other.  Dense random boards usually hold sixes of BOTH colours: the answer then depends on the scan
order, which is exactly what the known answers K-1 .. K-3 touch at a handful of positions only."""
import numpy as np
import pytest

import c6oracle as o

BACK = ((0, -1), (-1, 0), (-1, -1), (-1, 1))      # where the run that ends at a cell comes from: Right, Down, RightDown, LeftDown


def brute_force_winner(t):
    n = t.shape[0]
    for r in range(n):
        for c in range(n):
            p = t[r, c]
            if p == 0:
                continue
            for dr, dc in BACK:
                k, rr, cc = 1, r + dr, c + dc
                while 0 <= rr < n and 0 <= cc < n and t[rr, cc] == p:
                    k += 1; rr += dr; cc += dc
                if k >= 6:
                    return int(p)
    return 0


def all_sixes(t):
    """colours that have a run of six anywhere (order-free): used to count how often the scan order decides"""
    n, found = t.shape[0], set()
    for r in range(n):
        for c in range(n):
            p = t[r, c]
            if p == 0:
                continue
            for dr, dc in BACK:
                rr, cc, k = r, c, 0
                while 0 <= rr < n and 0 <= cc < n and t[rr, cc] == p:
                    k += 1; rr += dr; cc += dc
                if k >= 6:
                    found.add(int(p))
    return found


@pytest.mark.parametrize("density", [0.35, 0.6, 0.85, 1.0])
def test_random_boards(n, density):
    rng = np.random.default_rng(1000 * n + int(100 * density))
    both = wins = 0
    for _ in range(120 if n == 19 else 200):
        t = rng.choice(np.array([o.BLACK, o.WHITE], np.int8), size=(n, n))
        t[rng.random((n, n)) >= density] = 0
        want = brute_force_winner(t)
        assert o.search(t) == want, t
        wins += want != 0
        both += len(all_sixes(t)) == 2
    if density >= 0.85:
        assert wins > 0      # (dense boards do hold sixes ...
    if density == 1.0 and n >= 15:
        assert both > 0      #  ... and full ones of both colours: the scan order decided)


def test_planted_lines_in_noise(n):
    """one line of exactly six (or an overline of seven / eight) planted in sparse noise of the other colour and of its own, every direction and border"""
    rng = np.random.default_rng(77 + n)
    dirs = ((0, 1), (1, 0), (1, 1), (1, -1))
    for trial in range(300):
        t = np.zeros((n, n), np.int8)
        noise = rng.random((n, n)) < 0.12
        t[noise] = rng.choice(np.array([o.BLACK, o.WHITE], np.int8), size=int(noise.sum()))
        p = o.BLACK if trial & 1 else o.WHITE
        length = 6 + trial % 3
        dr, dc = dirs[trial % 4]
        r0 = int(rng.integers(0, n - (length - 1) * dr)) if dr else int(rng.integers(0, n))
        c0 = int(rng.integers(0, n - (length - 1))) if dc == 1 else (int(rng.integers(length - 1, n)) if dc == -1 else int(rng.integers(0, n)))
        for k in range(length):
            t[r0 + k * dr, c0 + k * dc] = p
        want = brute_force_winner(t)
        assert want != 0 and o.search(t) == want


def test_random_stone_sequences_against_a_python_model_of_game_set(n):
    """game_impl/mod.rs:125-169, restated as a few lines of Python: Black opens with ONE stone, then two per turn; a position outside the board is
    InvalidPosition, an occupied one AlreadySetPosition, and neither changes the game; SetResult carries the mover and the stones it has left
    BEFORE the turn switches.  Random sequences with both kinds of illegal stones mixed in, to the end of the board; the winner after every stone
    against the direct count above."""
    rng = np.random.default_rng(500 + n)
    for _ in range(6):
        g = o.new_game(n)
        board, turn, remain = np.zeros((n, n), np.int8), o.BLACK, 1
        res = o.SetResult()
        stones = 0
        while stones < n * n:
            kind = rng.random()
            if kind < 0.08:
                r, c = (int(rng.integers(n, n + 3)), int(rng.integers(0, n + 3))) if rng.random() < 0.5 else (int(rng.integers(0, n + 3)), int(rng.integers(n, n + 3)))
            elif kind < 0.2 and stones:
                rr, cc = np.nonzero(board)
                k = int(rng.integers(0, len(rr)))
                r, c = int(rr[k]), int(cc[k])
            else:
                rr, cc = np.nonzero(board == 0)
                k = int(rng.integers(0, len(rr)))
                r, c = int(rr[k]), int(cc[k])
            rc = o.lib().c6o_game_set(g, r, c, res)
            if r >= n or c >= n:
                assert rc == o.ERR_INVALID
            elif board[r, c] != 0:
                assert rc == o.ERR_ALREADY
            else:
                assert rc == o.OK
                board[r, c] = turn
                remain -= 1
                assert (res.player, res.num_remain, res.row, res.col) == (turn, remain, r, c)
                if remain <= 0:
                    remain, turn = 2, -turn
                stones += 1
            assert (g.turn, g.num_remain) == (turn, remain) and (g.np_board() == board).all()
            if stones % 7 == 0 or stones > n * n - 3:
                assert o.lib().c6o_game_is_end(g) == brute_force_winner(board)
