"""K-6: Connect6/libconnect6/src/policy/simulate/tests.rs:10-138 on the oracle."""
import ctypes as C

import c6oracle as o

B, Z, W = o.BLACK, o.NONE, o.WHITE


def test_from_game(n):  # simulate/tests.rs:10-36
    g = o.new_game(n)
    s = o.sim_from_game(g)
    assert (s.turn, s.num_remain) == (B, 1)
    assert s.possible_list() == list(range(n * n))
    o.game_set(g, 1, 2)
    s = o.sim_from_game(g)
    assert (s.turn, s.num_remain) == (W, 2)
    assert 1 * n + 2 not in s.possible_list() and s.npossible == n * n - 1


def test_validate(n):  # simulate/tests.rs:52-69
    L = o.lib()
    g = o.new_game(n)
    s = o.sim_from_game(g)
    assert L.c6o_sim_validate(C.byref(s), 0, 0)
    L.c6o_sim_simulate_in(C.byref(s), 0, 0)
    assert not L.c6o_sim_validate(C.byref(s), 0, 0)
    for row, col in ((1, 40), (n, 0), (0, n), (n - 1, n - 1), (3, 2)):
        assert bool(L.c6o_sim_validate(C.byref(s), row, col)) == (row < n and col < n)


def test_next_turn(n):  # simulate/tests.rs:71-81
    L = o.lib()
    s = o.sim_from_game(o.new_game(n))
    assert L.c6o_sim_next_turn(C.byref(s)) == W
    L.c6o_sim_simulate_in(C.byref(s), 0, 0)
    assert L.c6o_sim_next_turn(C.byref(s)) == W
    L.c6o_sim_simulate_in(C.byref(s), 0, 1)
    assert L.c6o_sim_next_turn(C.byref(s)) == B


def test_simulate_in_and_rollback(n):  # simulate/tests.rs:104-138
    L = o.lib()
    s = o.sim_from_game(o.new_game(n))
    L.c6o_sim_simulate_in(C.byref(s), 0, 0)
    assert s.np_board()[0, 0] == B
    assert (s.turn, s.num_remain) == (W, 2)
    assert 0 not in s.possible_list()
    L.c6o_sim_rollback_in(C.byref(s), 0, 0)
    assert (s.turn, s.num_remain) == (B, 1)
    assert not s.np_board().any()
    assert s.npossible == n * n
    # rollback appends at the END of `possible` (simulate/mod.rs:294)
    assert s.possible_list()[-1] == 0


def test_rollback_turn_bookkeeping(n):  # simulate/mod.rs:296-300 inverse of :268-273
    L = o.lib()
    s = o.sim_from_game(o.new_game(n))
    states = []
    for c in range(6):
        states.append((s.turn, s.num_remain))
        L.c6o_sim_simulate_in(C.byref(s), 1, c)
    for c in reversed(range(6)):
        L.c6o_sim_rollback_in(C.byref(s), 1, c)
        assert (s.turn, s.num_remain) == states[c]
