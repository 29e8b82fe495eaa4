"""K1 board kernel (HIP, through the C ABI) against the oracle: bit-exact.
Reference: game/game_impl/mod.rs:148-169, search_winner/mod.rs:201-246 and their tests."""
import ctypes as C

import numpy as np
import pytest

import c6oracle as o
from gpu_common import need_gpu, pkg

pytestmark = pytest.mark.gpu
B, Z, W = -1, 0, 1


def test_search_known_answers(n):  # search_winner/tests.rs:102-216 through the GPU scan
    need_gpu()
    boards = []
    t = np.zeros((n, n), np.int8)
    t[0, 0:5] = B; boards.append(t.copy())
    t[0, 5] = B; boards.append(t.copy())
    for i in range(5): t[5 - i, 0] = W
    boards.append(t.copy())
    t[0, 0] = W; boards.append(t.copy())
    for i in range(5): t[5 - i, 5 - i] = B
    boards.append(t.copy())
    t[0, 0] = B; boards.append(t.copy())
    for i in range(5): t[5 - i, i] = W
    boards.append(t.copy())
    t[0, 5] = W; boards.append(t.copy())
    expect = [Z, B, B, W, W, B, B, W]
    for base in (0, n - 1):
        for init in (0, 1, n - 6):
            t = np.zeros((n, n), np.int8); t[base, init:init + 5] = W; boards.append(t.copy()); expect.append(Z)
            t[base, init + 5] = W; boards.append(t.copy()); expect.append(W)
            t = np.zeros((n, n), np.int8); t[init:init + 5, base] = B; boards.append(t.copy()); expect.append(Z)
            t[init + 5, base] = B; boards.append(t.copy()); expect.append(B)
    for init in (0, 1, n - 6):
        t = np.zeros((n, n), np.int8)
        for i in range(5): t[init + i, init + i] = W
        boards.append(t.copy()); expect.append(Z)
        t[init + 5, init + 5] = W; boards.append(t.copy()); expect.append(W)
        t = np.zeros((n, n), np.int8)
        for i in range(5): t[init + i, n - 1 - init - i] = B
        boards.append(t.copy()); expect.append(Z)
        t[init + 5, n - init - 6] = B; boards.append(t.copy()); expect.append(B)
    got = pkg.search_winner(n, pkg.to_soa(np.array(boards)))
    assert got.tolist() == expect
    assert [o.search(b) for b in boards] == expect


def test_search_random_boards_vs_oracle(n):
    """dense random boards: many simultaneous sixes of both colours -> the scan-order tie-break matters"""
    need_gpu()
    rng = np.random.default_rng(n)
    G = 3001  # ragged: not a multiple of 4
    boards = rng.choice(np.array([-1, 0, 1], np.int8), size=(G, n, n), p=[0.42, 0.16, 0.42])
    boards[::7] = rng.choice(np.array([-1, 0, 1], np.int8), size=boards[::7].shape, p=[0.2, 0.6, 0.2])
    got = pkg.search_winner(n, pkg.to_soa(boards))
    want = np.array([o.search(b) for b in boards], np.int8)
    assert (got == want).all()
    assert len(set(want.tolist())) == 3


def test_random_playouts_vs_oracle(n):
    """G games of uniformly random legal play, stone by stone: board, turn, remain, winner identical"""
    need_gpu()
    rng = np.random.default_rng(100 + n)
    G = 257
    soa = np.zeros((n * n, G), np.int8)
    turn = np.full(G, B, np.int8)
    remain = np.ones(G, np.int8)
    games = [o.new_game(n) for _ in range(G)]
    done = np.zeros(G, bool)
    for step in range(n * n):
        pos = np.full(G, -1, np.int32)
        for g in range(G):
            if done[g]:
                continue
            empt = np.flatnonzero(soa[:, g] == 0)
            if empt.size == 0:
                done[g] = True
                continue
            pos[g] = rng.choice(empt)
        if (pos < 0).all():
            break
        soa, turn, remain, win, err = pkg.board_step(n, soa, turn, remain, pos)
        for g in range(G):
            if pos[g] < 0:
                continue
            e, _ = o.game_set(games[g], int(pos[g]) // n, int(pos[g]) % n)
            assert e == o.OK and err[g] == 0
            assert (games[g].np_board().reshape(-1) == soa[:, g]).all()
            assert (games[g].turn, games[g].num_remain) == (turn[g], remain[g])
            w = o.lib().c6o_game_is_end(C.byref(games[g]))
            assert w == win[g]
            if w != 0:
                done[g] = True
    assert done.sum() > G // 2


def test_set_errors(n):  # game_impl/tests.rs:30-38 error cases
    need_gpu()
    soa = np.zeros((n * n, 6), np.int8)
    soa[0, 1] = B
    turn = np.array([B, W, W, B, B, W], np.int8)
    remain = np.array([1, 2, 2, 1, 1, 2], np.int8)
    pos = np.array([0, 0, n * n, -1, n * n + 5, 3], np.int32)
    soa2, t2, r2, win, err = pkg.board_step(n, soa, turn, remain, pos)
    assert err.tolist() == [0, pkg.ERR_ALREADY_SET, pkg.ERR_INVALID_POSITION, 0, pkg.ERR_INVALID_POSITION, 0]
    assert soa2[0, 0] == B and (t2[0], r2[0]) == (W, 2)        # first stone: SetResult{Black,0}, then White/2
    assert soa2[0, 1] == B and (t2[1], r2[1]) == (W, 2)        # rejected: unchanged
    assert soa2[3, 5] == W and (t2[5], r2[5]) == (W, 1)
    assert (soa2[:, 2] == 0).all() and (soa2[:, 3] == 0).all()


def test_legal_mask(n):
    need_gpu()
    rng = np.random.default_rng(7)
    G = 130
    boards = rng.integers(-1, 2, size=(G, n * n)).astype(np.int8)
    m = pkg.legal_mask(n, pkg.to_soa(boards))
    bits = np.unpackbits(m, axis=0, bitorder="little")[: n * n].T
    assert (bits == (boards == 0)).all()
