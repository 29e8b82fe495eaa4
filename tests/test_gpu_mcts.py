"""K2 tree kernels: whole self-play games on the GPU against the seeded CPU oracle -- moves, winners and
root visit counts bit-identical (the random stream is unpinned at the reference, parity for MCTS
is exact agreement GPU <-> seeded oracle plus the K-8 invariants)."""
import os

import numpy as np
import pytest

import c6oracle as o
from gpu_common import need_gpu, pkg

pytestmark = pytest.mark.gpu


def run_pair(n, G, S, seed, evaluator, oracle_kind, **kw):
    eng = pkg.Engine(board_size=n, num_games=G, num_simulation=S, seed=seed, evaluator=evaluator, **kw)
    eng.enable_tuples(G * n * n)
    eng.run()
    st = eng.stats()
    games = {g["uid"]: g for g in eng.fetch_games()}
    tup = eng.fetch_tuples(G * n * n)
    eng.close()
    ref = o.self_play(n, num=G, threads=min(G, os.cpu_count() or 1), num_simulation=S, seed=seed, eval_kind=oracle_kind,
                      want_pi=True, epsilon=kw.get("epsilon", 0.25), alpha=kw.get("dirichlet_alpha", 0.03),
                      c_puct=kw.get("c_puct", 1.0))
    return st, games, tup, ref


def check_games(n, G, games, ref):
    assert sorted(games) == list(range(G))
    for uid in range(G):
        mv = [r * n + c for _, _, (r, c) in ref[uid]["path"]]
        assert list(games[uid]["moves"]) == mv, (uid, list(games[uid]["moves"])[:12], mv[:12])
        assert games[uid]["winner"] == ref[uid]["winner"]
        if ref[uid]["winner"] != 0:
            assert ref[uid]["path"][-1][0] == ref[uid]["winner"]      # last.turn == winner, tests.rs:198-202


@pytest.mark.parametrize("n,G,S", [(9, 12, 24), (15, 6, 16), (19, 4, 12)])
def test_self_play_random_evaluator_bit_exact(n, G, S):
    need_gpu()
    st, games, tup, ref = run_pair(n, G, S, seed=77 + n, evaluator=pkg.EVAL_RANDOM, oracle_kind=o.EVAL_RANDOM)
    assert st["overflow_games"] == 0 and st["games_finished"] == G
    check_games(n, G, games, ref)
    # training tuples: (state before the move, pi = root child visits, turn, z = winner)
    want = {}
    for uid in range(G):
        for k, (t, b, pos) in enumerate(ref[uid]["path"]):
            want[b.tobytes() + bytes([k % 251])] = (t, ref[uid]["pi"][k], ref[uid]["winner"])
    assert tup["state"].shape[0] == sum(len(r["path"]) for r in ref)
    # group the emitted tuples back per game via replay
    by_state = {}
    for i in range(tup["state"].shape[0]):
        by_state.setdefault(tup["state"][i].tobytes(), []).append(i)
    for uid in range(G):
        for k, (t, b, pos) in enumerate(ref[uid]["path"]):
            idxs = by_state[b.tobytes()]
            assert any(tup["turn"][i] == t and tup["z"][i] == ref[uid]["winner"] and
                       (tup["pi"][i].astype(np.int64) == ref[uid]["pi"][k]).all() for i in idxs), (uid, k)


def test_self_play_uniform_evaluator_bit_exact():
    need_gpu()
    n, G, S = 9, 8, 40
    st, games, tup, ref = run_pair(n, G, S, seed=5, evaluator=pkg.EVAL_UNIFORM, oracle_kind=o.EVAL_UNIFORM)
    assert st["overflow_games"] == 0
    check_games(n, G, games, ref)


def test_hyperparameters_are_honoured():
    need_gpu()
    n, G, S = 9, 6, 20
    st, games, tup, ref = run_pair(n, G, S, seed=9, evaluator=pkg.EVAL_RANDOM, oracle_kind=o.EVAL_RANDOM,
                                   epsilon=0.4, dirichlet_alpha=0.3, c_puct=2.5)
    check_games(n, G, games, ref)


def test_k8_invariants_after_searches(n):
    """alphazero_policy/tests.rs:42-183 on the engine: after 1 search the root has visit 1 and N*N children with
    visit 0 whose n_prob is the root's prob; after 2 searches exactly one child carries a value and
    root.q_value * 2 == child.value."""
    need_gpu()
    eng = pkg.Engine(board_size=n, num_games=3, num_simulation=1000, seed=3, evaluator=pkg.EVAL_RANDOM)
    eng.step(1)
    for g in range(3):
        r = eng.root(g)
        assert r["root_visit"] == 1 and r["root_value"] != 0 and r["root_q_value"] == 0
        assert (r["visits"] == 0).all() and np.count_nonzero(r["prior"]) > n * n // 2
    az = o.AlphaZero(n, eval_kind=o.EVAL_RANDOM, seed=3, game_uid=1)
    s = o.sim_from_game(o.new_game(n))
    az.init(s)
    az.search(s)
    assert (az.node_prob(np.zeros((n, n), np.int8)).reshape(-1) == eng.root(1)["prior"]).all()
    eng.step(1)
    for g in range(3):
        r = eng.root(g)
        assert r["root_visit"] == 2 and (r["visits"] > 0).sum() == 1 and r["visits"].sum() == 1
        assert r["root_q_value"] != 0
        assert np.float32(r["root_q_value"]) * np.float32(2) == np.float32(r["root_q_sum"])
    eng.close()


def test_tree_state_matches_oracle_mid_game():
    """root statistics (child visits, q, priors) after every move of one game equal the oracle's"""
    need_gpu()
    n, S, seed = 9, 32, 11
    eng = pkg.Engine(board_size=n, num_games=2, num_simulation=S, seed=seed, evaluator=pkg.EVAL_RANDOM)
    az = o.AlphaZero(n, num_simulation=S, eval_kind=o.EVAL_RANDOM, seed=seed, game_uid=1)
    g = o.new_game(n)
    for move in range(14):
        eng.step(S - 1)
        s = o.sim_from_game(g)
        az.init(s)
        for _ in range(S - 1):
            az.search(s)
        r = eng.root(1)
        vis, q, p = az.children(g.np_board())
        assert (r["board"] == g.np_board()).all()
        empty = g.np_board().reshape(-1) == 0
        assert (r["visits"][empty] == vis[empty]).all(), move
        assert (r["q"][empty] == q[empty]).all(), move
        assert (r["prior"][empty] == p[empty]).all(), move
        eng.step(1)
        az.search(s)
        cell = az.policy(s)
        az.prune(s)
        o.game_set(g, cell // n, cell % n)
        import ctypes as C
        if o.lib().c6o_game_is_end(C.byref(g)) != 0:
            break
    eng.close()


def test_recycle_and_stagger_keep_games_exact():
    """steady-state mode: slots restart with uid += stride; staggered phases do not change any game"""
    need_gpu()
    n, G, S, seed = 9, 4, 10, 21
    eng = pkg.Engine(board_size=n, num_games=G, num_simulation=S, seed=seed, evaluator=pkg.EVAL_RANDOM, recycle=True)
    got = []
    while len(got) < 3 * G:
        eng.step(S * 8)
        got += eng.fetch_games()
    eng.close()
    uids = sorted(g["uid"] for g in got)
    assert uids[:G] == list(range(G)) and set(uids) <= set(range(0, 100 * G))
    ref = o.self_play(n, num=max(uids) + 1, threads=os.cpu_count() or 1, num_simulation=S, seed=seed)
    for g in got:
        mv = [r * n + c for _, _, (r, c) in ref[g["uid"]]["path"]]
        assert list(g["moves"]) == mv and g["winner"] == ref[g["uid"]]["winner"]


def test_tables_beyond_the_lds_tag_row_stay_exact():
    """Trees that keep more than 4096 records alive -> tables of 16384+ slots: the tag row (one byte per slot) no longer fits the 8192
    bytes a wave keeps in LDS and is read in memory through the chunk list instead (tree_kernels.hip: C6Tags, C6_TAG_LDS_MAX).  The
    stress evaluator (a deep tree: ~6 x num_simulation records) at 1200 simulations per move, in the fused search kernel (device
    evaluator) and in the split select / expand kernels (the same evaluator as a host callback), against the oracle's unbounded map."""
    need_gpu()
    n, G, S, eps = 9, 3, 1200, 0.001

    def oracle_stress(ctx, turn, nn, ln, boards, values, policies):
        b = np.ctypeslib.as_array(boards, shape=(ln, nn * nn))
        v, p = pkg.stress_eval(nn, turn, b)
        np.ctypeslib.as_array(values, shape=(ln,))[:] = v
        np.ctypeslib.as_array(policies, shape=(ln, nn * nn))[:] = p
        return 0

    for seed, evaluator in ((21, pkg.EVAL_STRESS), (22, pkg.EVAL_CALLBACK)):
        eng = pkg.Engine(board_size=n, num_games=G, num_simulation=S, seed=seed, epsilon=eps, evaluator=evaluator)
        if evaluator == pkg.EVAL_CALLBACK:
            eng.set_callback(lambda player, boards: pkg.stress_eval(n, player, boards))
        eng.run()
        st = eng.stats()
        games = {g["uid"]: g for g in eng.fetch_games()}
        eng.close()
        assert st["overflow_games"] == 0 and st["max_records_used"] > 4096 + S, st      # 2 x (live + S) > 8192 slots: tags in memory
        ref = o.self_play(n, num=G, threads=1, num_simulation=S, seed=seed, epsilon=eps, eval_kind=o.EVAL_CUSTOM, eval_fn=oracle_stress)
        check_games(n, G, games, ref)


def test_fetch_games_in_pieces_hands_out_every_game_once():
    """c6_engine_fetch_games with room for fewer games than are queued (r5): every call returns the NEXT games -- before, a caller whose
    `cap` was smaller than the queue was handed the first `cap` games again at every call.  The pieces together are the single full fetch."""
    need_gpu()
    n, G, S, seed = 9, 7, 8, 99
    def play():
        eng = pkg.Engine(board_size=n, num_games=G, num_simulation=S, seed=seed, evaluator=pkg.EVAL_RANDOM)
        eng.run()
        return eng
    e1 = play()
    whole = e1.fetch_games()
    assert len(whole) == G and e1.fetch_games() == []
    e1.close()
    e2 = play()
    pieces = []
    for cap in (3, 3, 3, 3):
        got = e2.fetch_games(cap=cap)
        assert len(got) <= cap
        pieces += got
    assert e2.fetch_games(cap=3) == []
    e2.close()
    assert len(pieces) == G and sorted(g["uid"] for g in pieces) == list(range(G))
    key = lambda g: g["uid"]
    for a, b in zip(sorted(whole, key=key), sorted(pieces, key=key)):
        assert a["winner"] == b["winner"] and list(a["moves"]) == list(b["moves"])


def test_fetch_tuples_in_pieces_keeps_the_rest():
    """c6_engine_fetch_tuples with room for fewer records than the buffer holds (r5): the rest stays for the next call (the reset used to drop
    it); the pieces in order are the single full fetch."""
    need_gpu()
    n, G, S, seed = 9, 4, 8, 5
    def play():
        eng = pkg.Engine(board_size=n, num_games=G, num_simulation=S, seed=seed, evaluator=pkg.EVAL_RANDOM)
        eng.enable_tuples(G * n * n)
        eng.run()
        return eng
    e1 = play()
    whole = e1.fetch_tuples(G * n * n)
    total = whole["state"].shape[0]
    assert total > 3 * 17 and e1.fetch_tuples(8)["state"].shape[0] == 0
    e1.close()
    e2 = play()
    parts = []
    while True:
        t = e2.fetch_tuples(17)
        if t["state"].shape[0] == 0:
            break
        assert t["state"].shape[0] <= 17
        parts.append(t)
    e2.close()
    assert sum(p["state"].shape[0] for p in parts) == total
    for k in whole:
        assert np.array_equal(np.concatenate([p[k] for p in parts], axis=0), whole[k]), k
