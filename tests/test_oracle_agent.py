"""K-5 and the agent error paths: Connect6/libconnect6/src/agent/agent_impl/tests.rs:36-130,
plus the deterministic cpp_play vectors derived from test_cppbind/main.cpp:31-44."""
import numpy as np
import pytest

import c6oracle as o

B, Z, W = o.BLACK, o.NONE, o.WHITE


def scripted(moves):
    it = iter(moves)
    return lambda g: next(it)


def test_play_invalid_position(n):  # agent_impl/tests.rs:36-48
    r = o.agent_play(n, scripted([(n, n)]))
    assert o.lib().c6o_strerror(r["err"]) == b"invalid position"


def test_play_already_set_position(n):  # agent_impl/tests.rs:50-63
    r = o.agent_play(n, scripted([(0, 0), (0, 0)]))
    assert o.lib().c6o_strerror(r["err"]) == b"already set position"


def test_play_scripted_game(n):  # agent_impl/tests.rs:65-130
    record = [(0, 0), (0, 1), (0, 2), (1, 0), (2, 0), (0, 3), (0, 4), (3, 0), (4, 0), (0, 5), (0, 6)]
    r = o.agent_play(n, scripted(record))
    assert r["err"] == o.OK and r["winner"] == W and len(r["path"]) == 11
    turn, remain = B, 1
    board = np.zeros((n, n), np.int8)
    prev = None
    for k, (t, b, pos) in enumerate(r["path"]):
        if prev is not None:
            board[prev] = turn
            remain -= 1
            if remain <= 0:
                remain, turn = 2, -turn
        assert t == turn and (b == board).all() and pos == record[k]
        prev = pos


def test_policy_none_ends_as_draw(n):  # agent_impl/mod.rs:135-137
    r = o.agent_play(n, scripted([(0, 0), None]))
    assert r["winner"] == Z and len(r["path"]) == 1


def last_empty(g):  # test_cppbind/main.cpp:31-44 main_policy: LAST empty cell
    b = g.np_board().reshape(-1)
    idx = np.flatnonzero(b == 0)
    return None if idx.size == 0 else (int(idx[-1]) // g.n, int(idx[-1]) % g.n)


def first_empty(g):
    b = g.np_board().reshape(-1)
    idx = np.flatnonzero(b == 0)
    return None if idx.size == 0 else (int(idx[0]) // g.n, int(idx[0]) % g.n)


# SURVEY 8c "derived deterministic end-to-end vectors" (scratch there, re-derived here by the oracle)
@pytest.mark.parametrize("size,policy,winner,stones,last", [
    (15, last_empty, B, 81, (9, 9)), (19, last_empty, B, 101, (13, 13)), (9, last_empty, W, 46, (3, 8)),
    (15, first_empty, B, 81, (5, 5)), (19, first_empty, B, 101, (5, 5)), (9, first_empty, W, 46, (5, 0)),
])
def test_deterministic_policy_games(size, policy, winner, stones, last):
    r = o.agent_play(size, policy)
    assert r["winner"] == winner and len(r["path"]) == stones and r["path"][-1][2] == last
    assert r["path"][-1][0] == winner  # last.turn == winner (alphazero_policy/tests.rs:198-202)
