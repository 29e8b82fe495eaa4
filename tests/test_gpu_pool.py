"""The per-game record pool against the reference's unbounded HashMap<u64, Node> (alphazero_policy/mod.rs:186-190,360-377):
growth is invisible in the results, exhaustion is never silent."""
import ctypes as C
import os

import numpy as np
import pytest

import c6oracle as o
from gpu_common import need_gpu, pkg

pytestmark = pytest.mark.gpu
N = 9


def peaked(player, boards, n=N):
    """A prior of e^(-14 d^2) around the centre-most empty cell (invariant under the 8 dihedral images, so the averaged prior
    keeps the peak; every other child is ~1e-6) and a value every mover likes two plies up (-0.9 * side to move: with
    two-stone turns the mover two plies above a leaf is always the other colour): with epsilon = 0.001 nearly every search
    follows the peaked child level after level -- the select depth reaches the teens (mean 6.5 on the oracle)"""
    b = np.asarray(boards).reshape(-1, n * n)
    rr, cc = np.divmod(np.arange(n * n), n)
    d2 = ((rr - n // 2) ** 2 + (cc - n // 2) ** 2).astype(np.float32)
    logits = np.where(b == 0, -14.0 * d2[None, :], -1e9).astype(np.float32)
    e = np.exp(logits - logits.max(axis=1, keepdims=True))
    pol = e / e.sum(axis=1, keepdims=True)
    return np.full(b.shape[0], -0.9 * player, np.float32), pol.astype(np.float32)


def oracle_eval(ctx, turn, nn, ln, boards, values, policies):
    b = np.ctypeslib.as_array(boards, shape=(ln, nn * nn))
    v, p = peaked(turn, b)
    np.ctypeslib.as_array(values, shape=(ln,))[:] = v
    np.ctypeslib.as_array(policies, shape=(ln, nn * nn))[:] = p
    return 0


def oracle_games(G, S, seed, eps):
    ref = o.self_play(N, num=G, threads=1, num_simulation=S, seed=seed, epsilon=eps, eval_kind=o.EVAL_CUSTOM, eval_fn=oracle_eval)
    return [([r * N + c for _, _, (r, c) in g["path"]], g["winner"]) for g in ref]


def test_pool_grows_under_a_peaked_prior_and_games_stay_exact():
    """8 records per game to start with, a prior that drives the select depth far beyond 8: c6_engine_run doubles the pools (and
    the select-path buffer) as often as needed and every game still equals the oracle's unbounded map stone for stone."""
    need_gpu()
    G, S, seed, eps = 4, 60, 5, 0.001
    eng = pkg.Engine(board_size=N, num_games=G, num_simulation=S, seed=seed, epsilon=eps, evaluator=pkg.EVAL_CALLBACK, records_per_game=8)
    eng.set_callback(peaked)
    eng.run()
    st = eng.stats()
    games = {g["uid"]: g for g in eng.fetch_games()}
    eng.close()
    assert st["overflow_games"] == 0 and st["abandoned_games"] == 0 and len(games) == G
    assert st["pool_grows"] >= 3 and st["records_per_game"] >= 64 and st["max_depth"] > 8, st
    for uid, (mv, w) in enumerate(oracle_games(G, S, seed, eps)):
        assert list(games[uid]["moves"]) == mv and games[uid]["winner"] == w


def test_step_without_maintain_parks_the_game_and_maintain_resumes_it_exactly():
    """c6_engine_step never synchronises, so it cannot grow: the game waits in status overflow (visible in the stats), loses
    nothing, and continues after c6_engine_maintain -- same final game as the oracle."""
    need_gpu()
    G, S, seed, eps = 2, 30, 9, 0.001
    eng = pkg.Engine(board_size=N, num_games=G, num_simulation=S, seed=seed, epsilon=eps, evaluator=pkg.EVAL_CALLBACK, records_per_game=8)
    eng.set_callback(peaked)
    eng.step(20)
    st = eng.stats()
    assert st["overflow_games"] == G and st["live_games"] == 0 and st["simulations"] == 8 * G      # stopped when the 8 records were used
    eng.maintain()
    st = eng.stats()
    assert st["overflow_games"] == 0 and st["live_games"] == G and st["records_per_game"] == 16
    eng.run()
    games = {g["uid"]: g for g in eng.fetch_games()}
    eng.close()
    for uid, (mv, w) in enumerate(oracle_games(G, S, seed, eps)):
        assert list(games[uid]["moves"]) == mv and games[uid]["winner"] == w


def test_exhausted_pool_returns_partial_games_and_says_so():
    """growth capped (max_records_per_game): c6_engine_run finishes what it can, returns C6_ERR_OVERFLOW with a message and
    hands the stopped games back as partial results (winner 0, a prefix of the oracle's game) -- never C6_OK with games
    missing, never an abort."""
    need_gpu()
    G, S, seed, eps = 3, 40, 5, 0.001
    eng = pkg.Engine(board_size=N, num_games=G, num_simulation=S, seed=seed, epsilon=eps, evaluator=pkg.EVAL_CALLBACK, records_per_game=8,
                     max_records_per_game=32)
    eng.set_callback(peaked)
    with pytest.raises(pkg.C6Error) as ei:
        eng.run()
    assert "partial" in str(ei.value)
    st = eng.stats()
    games = {g["uid"]: g for g in eng.fetch_games()}
    eng.close()
    assert len(games) == G and st["abandoned_games"] >= 1 and st["overflow_games"] == st["abandoned_games"]
    ref = oracle_games(G, S, seed, eps)
    partial = 0
    for uid, (mv, w) in enumerate(ref):
        got = list(games[uid]["moves"])
        if len(got) < len(mv):
            partial += 1
            assert games[uid]["winner"] == 0 and got == mv[:len(got)]
        else:
            assert got == mv and games[uid]["winner"] == w
    assert partial == st["abandoned_games"]


def test_dropin_self_play_survives_an_exhausted_pool(monkeypatch, capfd):
    """cpp_self_play with the pool capped by the environment knobs: no abort; every game comes back, the stopped ones as
    {winner 0, stones so far} with the reason on stderr (SURVEY 8b "Errors")."""
    need_gpu()
    monkeypatch.setenv("C6_RECORDS", "48")          # the first move's 40 searches fit, the second move's do not
    monkeypatch.setenv("C6_MAX_RECORDS", "48")
    monkeypatch.setenv("C6_SEED", "3")
    n = pkg.REF_N

    def pol(turn, boards):
        return peaked(turn, boards, n)

    param = pkg.default_param()
    param.update(num_simulation=40, epsilon=0.001, num_game_thread=2)
    res = pkg.self_play(pol, param)
    err = capfd.readouterr().err
    assert len(res) == 2 and "partial" in err
    for winner, path in res:
        assert winner == 0 and 1 <= len(path) < n * n
