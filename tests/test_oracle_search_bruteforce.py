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
