"""Pins the oracle's game logic to the reference's own known-answer tests.

K-1/K-2/K-3: Connect6/libconnect6/src/game/search_winner/tests.rs:101-216
K-4:         Connect6/libconnect6/src/game/game_impl/tests.rs:4-52
Player:      Connect6/libconnect6/src/game/player/tests.rs:4-28
All are written in terms of BOARD_SIZE in the reference, so they are
instantiated at N = 9, 15 (the reference's compiled size) and 19.
"""
import numpy as np

import c6oracle as o

B, Z, W = o.BLACK, o.NONE, o.WHITE


def table(n):
    return np.zeros((n, n), np.int8)


def test_player_switch():  # player/tests.rs:4-28
    L = o.lib()
    assert L.c6o_player_switch(B) == W
    assert L.c6o_player_switch(W) == B
    assert L.c6o_player_switch(Z) == Z


def test_base_search(n):  # search_winner/tests.rs:102-139
    t = table(n)
    t[0, 0:5] = B
    assert o.search(t) == Z
    t[0, 5] = B
    assert o.search(t) == B
    for i in range(5):
        t[5 - i, 0] = W
    assert o.search(t) == B
    t[0, 0] = W
    assert o.search(t) == W
    for i in range(5):
        t[5 - i, 5 - i] = B
    assert o.search(t) == W
    t[0, 0] = B
    assert o.search(t) == B
    for i in range(5):
        t[5 - i, i] = W
    assert o.search(t) == B
    t[0, 5] = W
    assert o.search(t) == W


def test_variadic_boundary_search(n):  # search_winner/tests.rs:141-182
    for base in (0, n - 1):
        for init in (0, 1, n - 6):
            t = table(n)
            t[base, init:init + 5] = W
            assert o.search(t) == Z
            t[base, init + 5] = W
            assert o.search(t) == W
            t = table(n)
            t[init:init + 5, base] = B
            assert o.search(t) == Z
            t[init + 5, base] = B
            assert o.search(t) == B


def test_variadic_cross_search(n):  # search_winner/tests.rs:184-216
    for init in (0, 1, n - 6):
        t = table(n)
        for i in range(5):
            t[init + i, init + i] = W
        assert o.search(t) == Z
        t[init + 5, init + 5] = W
        assert o.search(t) == W
        t = table(n)
        for i in range(5):
            t[init + i, n - 1 - init - i] = B
        assert o.search(t) == Z
        t[init + 5, n - init - 6] = B
        assert o.search(t) == B


def test_overline_and_scan_order(n):
    # overlines count; with two sixes the row-major scan order decides
    t = table(n)
    t[2, 0:7] = W
    assert o.search(t) == W
    t = table(n)
    t[0:6, 0] = B        # black six completes at (5, 0)
    t[3, 1:7] = W        # white six completes at (3, 6): earlier in the scan
    assert o.search(t) == W
    t = table(n)
    t[0:6, 0] = B        # completes at (5,0)
    t[5, 1:7] = W        # completes at (5,6): same row, later column
    assert o.search(t) == B


def test_game_new(n):  # game_impl/tests.rs:4-12
    g = o.new_game(n)
    assert (g.turn, g.num_remain) == (B, 1)
    assert not g.np_board().any()


def test_game_play(n):  # game_impl/tests.rs:15-39
    g = o.new_game(n)
    err, r = o.game_set(g, 0, 0)
    assert err == o.OK
    assert (r.player, r.num_remain, r.row, r.col) == (B, 0, 0, 0)
    assert (g.turn, g.num_remain) == (W, 2)
    assert g.np_board()[0, 0] == B
    err, _ = o.game_set(g, 0, 0)
    assert o.lib().c6o_strerror(err) == b"already set position"
    err, _ = o.game_set(g, n, n)
    assert o.lib().c6o_strerror(err) == b"invalid position"


def test_game_is_end(n):  # game_impl/tests.rs:48-52
    import ctypes as C
    g = o.new_game(n)
    assert o.lib().c6o_game_is_end(C.byref(g)) == Z


def test_turn_sequence(n):  # game_impl/mod.rs:158-168: one stone, then two per turn
    g = o.new_game(n)
    seen = []
    for c in range(7):
        seen.append((g.turn, g.num_remain))
        o.game_set(g, 0, c)
    assert seen == [(B, 1), (W, 2), (W, 1), (B, 2), (B, 1), (W, 2), (W, 1)]
