"""Evaluation matches on the device (c6_engine_play_match): MultiPolicy(AlphaZero, RandomPolicy / DefaultPolicy) of
policy/multi_policy/mod.rs:61-71 with the reference's two other policies (random_policy/mod.rs:39-47, default_policy/mod.rs:186-246)
as engine-side movers.  The glue is checked against the same match driven stone by stone through the host (c6_engine_step +
c6_engine_apply_moves, each validated against the oracle elsewhere): identical games, stone for stone."""
import numpy as np
import pytest

import c6oracle as o
from gpu_common import need_gpu, pkg

pytestmark = pytest.mark.gpu


def host_match(n, G, S, seed, colour, opponent_moves):
    """the match through the host: the engine moves on its colour's turns, `opponent_moves(boards, live, stones)` elsewhere"""
    eng = pkg.Engine(board_size=n, num_games=G, num_simulation=S, seed=seed, evaluator=pkg.EVAL_RANDOM)
    boards = np.zeros((G, n * n), np.int8)
    live = np.ones(G, bool)
    turn, remain, stones = -1, 1, 0
    while live.any():
        if turn == colour:
            eng.step(S)
            pos = eng.last_moves()
        else:
            pos = opponent_moves(boards, live, stones)
            eng.apply_moves(np.where(live, pos, -2))
        moved = live & (pos >= 0)
        boards[moved, pos[moved]] = turn
        w = pkg.search_winner(n, pkg.to_soa(boards))
        live &= (pos >= 0) & (w == 0)
        stones += 1
        remain -= 1
        if remain <= 0:
            remain, turn = 2, -turn
    games = {g["uid"]: g for g in eng.fetch_games()}
    eng.close()
    return games


def same_games(a, b, G):
    assert sorted(a) == sorted(b) == list(range(G))
    for uid in range(G):
        assert list(a[uid]["moves"]) == list(b[uid]["moves"]) and a[uid]["winner"] == b[uid]["winner"], uid


@pytest.mark.parametrize("colour", [-1, 1])
def test_match_against_random_policy_on_device(colour):
    need_gpu()
    n, G, S, seed, opp_seed = 9, 8, 16, 3, 0x1234567800000009

    def random_policy(boards, live, stones):      # C6RNG tag 8 (connect6_amd.h): the k-th empty cell, k = (w * len) >> 32
        pos = np.full(G, -1, np.int32)
        for g in range(G):
            if live[g]:
                empty = np.flatnonzero(boards[g] == 0)
                if empty.size:
                    w = o.philox([stones, 0, 0, 8], [opp_seed & 0xffffffff, ((opp_seed >> 32) + g) & 0xffffffff])[0]
                    pos[g] = empty[(w * empty.size) >> 32]
        return pos

    want = host_match(n, G, S, seed, colour, random_policy)
    eng = pkg.Engine(board_size=n, num_games=G, num_simulation=S, seed=seed, evaluator=pkg.EVAL_RANDOM)
    eng.play_match(colour, "random", opp_seed=opp_seed)
    got = {g["uid"]: g for g in eng.fetch_games()}
    st = eng.stats()
    eng.close()
    assert st["live_games"] == 0 and st["overflow_games"] == 0
    same_games(got, want, G)
    for g in got.values():      # every stone legal, the winner is K1's
        assert len(set(g["moves"].tolist())) == len(g["moves"])


def test_match_against_default_policy_on_device():
    need_gpu()
    n, G, S, seed, iters = 9, 6, 12, 5, 20
    dp = pkg.DefaultPolicy(n, G, num_iter=iters, seed=seed + 1)
    want = host_match(n, G, S, seed, -1, lambda boards, live, stones: dp.next(boards, active=live.astype(np.uint8)))
    dp.close()
    eng = pkg.Engine(board_size=n, num_games=G, num_simulation=S, seed=seed, evaluator=pkg.EVAL_RANDOM)
    dp = pkg.DefaultPolicy(n, G, num_iter=iters, seed=seed + 1)
    eng.play_match(-1, "rollout", rollout=dp)
    got = {g["uid"]: g for g in eng.fetch_games()}
    eng.close()
    dp.close()
    same_games(got, want, G)


def test_play_match_rejects_bad_arguments():
    need_gpu()
    eng = pkg.Engine(board_size=9, num_games=2, num_simulation=4, evaluator=pkg.EVAL_RANDOM, recycle=True)
    with pytest.raises(pkg.C6Error, match="recycle"):
        eng.play_match(-1, "random")
    eng.close()
    eng = pkg.Engine(board_size=9, num_games=2, num_simulation=4, evaluator=pkg.EVAL_RANDOM)
    with pytest.raises(pkg.C6Error, match="c6_rollout"):
        eng.play_match(1, "rollout")
    with pytest.raises(pkg.C6Error, match="engine_colour"):
        eng.play_match(0, "random")
    eng.close()


def test_play_match_checks_that_the_engine_is_at_the_start():
    """the match loop tracks turn / remain on the host, so the precondition is checked: a second call (stones on the boards),
    or an engine that has already searched, is refused instead of playing the opponent's stones with the engine's search"""
    need_gpu()
    eng = pkg.Engine(board_size=9, num_games=3, num_simulation=6, seed=2, evaluator=pkg.EVAL_RANDOM)
    eng.play_match(-1, "random", opp_seed=5, max_stones=3)
    with pytest.raises(pkg.C6Error, match="start of its games"):
        eng.play_match(-1, "random", opp_seed=5)
    eng.close()
    eng = pkg.Engine(board_size=9, num_games=3, num_simulation=6, seed=2, evaluator=pkg.EVAL_RANDOM)
    eng.step(2)
    with pytest.raises(pkg.C6Error, match="start of its games"):
        eng.play_match(1, "random", opp_seed=5)
    eng.close()


@pytest.mark.parametrize("colour", [-1, 1])
def test_play_match_serves_parked_games_before_the_next_stone(colour):
    """a game whose tree outgrows the per-game caps is parked at its move boundary.  The match loop now raises the caps /
    grows the pool and lets the game ask again BEFORE the next stone, so every game stays in step: an engine created for 8 records
    per game (caps raised again and again under the stress evaluator's deep trees) plays exactly the games of an engine whose
    pool never runs short."""
    need_gpu()
    n, G, S, seed, eps, opp_seed = 9, 16, 60, 7, 0.001, 77
    res = []
    for rpg in (0, 8):
        eng = pkg.Engine(board_size=n, num_games=G, num_simulation=S, seed=seed, epsilon=eps, evaluator=pkg.EVAL_STRESS, stress_every=1, records_per_game=rpg)
        eng.play_match(colour, "random", opp_seed=opp_seed)
        st = eng.stats()
        assert st["live_games"] == 0 and st["overflow_games"] == 0 and st["parked_games"] == 0, st
        res.append(({g["uid"]: g for g in eng.fetch_games()}, st))
        eng.close()
    assert res[1][1]["cap_raises"] >= 1 and res[0][1]["cap_raises"] == 0, (res[0][1], res[1][1])
    same_games(res[0][0], res[1][0], G)
    # (inside play_match the caps are raised only for a game that IS parked -- serve_parked, engine.cpp -- so cap_raises >= 1 above
    #  means that games did wait at a move boundary and were resumed in step)
