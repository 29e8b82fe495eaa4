"""The reference's C ABI (cpp_self_play / cpp_play, N = 15) served by the GPU engine.
Reference tests mirrored: test_cppbind/main.cpp:46-57 (K-11) and the deterministic cpp_play vectors."""
import numpy as np
import pytest

import c6oracle as o
from gpu_common import need_gpu, pkg

pytestmark = pytest.mark.gpu
N = 15


def random_policy(turn, boards):  # Connect6/sample/python/random_policy.py:6-20 shape
    rng = np.random.default_rng(abs(hash(boards.tobytes())) % (2**32))
    n = boards.shape[0]
    return rng.uniform(-1, 1, n), rng.uniform(-1, 1, (n, N * N))


def check_result(res):
    for winner, path in res:
        assert len(path) > 0
        g = o.new_game(N)
        for turn, board, (r, c) in path:
            assert turn == g.turn and (board == g.np_board()).all()
            assert o.game_set(g, r, c)[0] == o.OK
        import ctypes as C
        assert o.lib().c6o_game_is_end(C.byref(g)) == winner
        if winner != 0:
            assert path[-1][0] == winner


def test_self_play_two_games():  # test_cppbind/main.cpp:46-52
    need_gpu()
    seen = []

    def policy(turn, boards):
        assert turn in (-1, 1) and boards.shape[0] % 8 == 0 and set(np.unique(boards)) <= {-1, 0, 1}
        seen.append(boards.shape[0])
        return random_policy(turn, boards)

    res = pkg.self_play(policy, {"num_simulation": 2, "num_game_thread": 2})
    assert len(res) == 2 and seen
    check_result(res)


def test_self_play_single_game_more_sims():
    need_gpu()
    res = pkg.self_play(random_policy, {"num_simulation": 12, "num_game_thread": 1})
    assert len(res) == 1
    check_result(res)


def test_play_main_policy():  # test_cppbind/main.cpp:31-44,54-57: last empty cell
    need_gpu()

    def main_policy(board):
        idx = np.flatnonzero(board.reshape(-1) == 0)
        return (int(idx[-1]) // N, int(idx[-1]) % N)

    res = pkg.play(main_policy, False, 2)
    assert len(res) == 2
    for winner, path in res:
        assert winner == -1 and len(path) == 81 and path[-1][2] == (9, 9)
    check_result(res)


def test_callback_evaluator_matches_oracle_with_same_callback():
    """same deterministic callback on both sides -> identical games (engine batches across games per side)"""
    need_gpu()
    n, G, S, seed = 9, 5, 14, 8

    def fn(player, boards):
        b = boards.astype(np.float32)
        k = np.arange(n * n, dtype=np.float32)
        pol = np.cos(b * 1.7 + k[None, :] * 0.37 + player) * 0.5
        val = np.tanh(b.sum(axis=1) * 0.1 + 0.05 * player)
        return val.astype(np.float32), pol.astype(np.float32)

    eng = pkg.Engine(board_size=n, num_games=G, num_simulation=S, seed=seed, evaluator=pkg.EVAL_CALLBACK)
    eng.set_callback(fn)
    eng.run()
    games = {g["uid"]: g for g in eng.fetch_games()}
    eng.close()

    def oracle_eval(ctx, turn, nn, ln, boards, values, policies):
        b = np.ctypeslib.as_array(boards, shape=(ln, nn * nn))
        v, p = fn(turn, b)
        np.ctypeslib.as_array(values, shape=(ln,))[:] = v
        np.ctypeslib.as_array(policies, shape=(ln, nn * nn))[:] = p
        return 0

    ref = o.self_play(n, num=G, threads=1, num_simulation=S, seed=seed, eval_kind=o.EVAL_CUSTOM, eval_fn=oracle_eval)
    for uid in range(G):
        mv = [r * n + c for _, _, (r, c) in ref[uid]["path"]]
        assert list(games[uid]["moves"]) == mv and games[uid]["winner"] == ref[uid]["winner"]
