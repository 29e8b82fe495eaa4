"""Parity at BASELINE.json's full sizes through size-independent properties, plus exact oracle checks on samples.
(configs: 4096 x 19x19 x 400 sims, 32768 x 9x9 x 100 sims.)"""
import ctypes as C
import os

import numpy as np
import pytest

import c6oracle as o
from gpu_common import need_gpu, pkg

pytestmark = pytest.mark.gpu


def test_board_kernel_32768_games_9x9():
    """config 5 shape: one stone on each of 32768 random mid-game boards: exactly one cell changes per game, turn/remain
    follow the 1-then-2 rule, winners agree with the oracle on a sample, and winner == 0 wherever no six exists."""
    need_gpu()
    n, G = 9, 32768
    rng = np.random.default_rng(5)
    boards = np.zeros((G, n * n), np.int8)
    stones = rng.integers(0, 60, G)
    for g in range(G):
        idx = rng.permutation(n * n)[: stones[g]]
        boards[g, idx] = rng.choice(np.array([-1, 1], np.int8), stones[g])
    turn = rng.choice(np.array([-1, 1], np.int8), G)
    remain = rng.integers(1, 3, G).astype(np.int8)
    pos = np.array([rng.choice(np.flatnonzero(boards[g] == 0)) for g in range(G)], np.int32)
    soa2, t2, r2, win, err = pkg.board_step(n, pkg.to_soa(boards), turn, remain, pos)
    after = soa2.T
    assert (err == 0).all()
    diff = after != boards
    assert (diff.sum(axis=1) == 1).all() and (after[np.arange(G), pos] == turn).all()
    assert ((r2 == 1) == (remain == 2)).all() and ((t2 == -turn) == (remain == 1)).all() and ((r2 == 2) == (remain == 1)).all()
    sample = rng.choice(G, 600, replace=False)
    for g in sample:
        assert o.search(after[g].reshape(n, n)) == win[g]
    # checksum of checksums: the full-scan kernel on the same boards gives the same winners
    assert (pkg.search_winner(n, soa2) == win).all()


def test_engine_4096_games_19x19_invariants_and_samples():
    """config 2 shape with the on-device RandomEvaluator: after R rounds every one of the 4096 roots has visit R and
    sum(child visits) == R - 1 (every search but the first passes through a child: alphazero_policy/tests.rs:90-136
    generalised), nothing overflowed; sampled games equal the seeded oracle bit for bit."""
    need_gpu()
    n, G, S, R, seed = 19, 4096, 400, 24, 99
    eng = pkg.Engine(board_size=n, num_games=G, num_simulation=S, seed=seed, evaluator=pkg.EVAL_RANDOM)
    eng.step(R)
    st = eng.stats()
    assert st["simulations"] == G * R and st["overflow_games"] == 0 and st["live_games"] == G
    assert 0.5 < st["select_steps"] / st["simulations"] < 3.0
    for g in list(range(0, G, 257)) + [G - 1]:
        r = eng.root(g)
        assert r["root_visit"] == R and r["visits"].sum() == R - 1 and r["stones"] == 0 and r["nsearch"] == R
    for g in (0, 1, 4095):
        az = o.AlphaZero(n, num_simulation=S, eval_kind=o.EVAL_RANDOM, seed=seed, game_uid=g)
        s = o.sim_from_game(o.new_game(n))
        az.init(s)
        for _ in range(R):
            az.search(s)
        vis, q, p = az.children(np.zeros((n, n), np.int8))
        r = eng.root(g)
        assert (r["visits"] == vis).all() and (r["q"] == q).all() and (r["prior"] == p).all()
    eng.close()


def test_tuples_round_trip_4096_games():
    """encode -> decode round trip of the training tuples at 4096 games: every emitted state is the replay of its game's
    move list, z is the game's winner and turn follows the 1-then-2 rule."""
    need_gpu()
    n, G, S = 9, 4096, 3
    eng = pkg.Engine(board_size=n, num_games=G, num_simulation=S, seed=1, evaluator=pkg.EVAL_UNIFORM)
    eng.enable_tuples(G * n * n)
    eng.run()
    games = eng.fetch_games()
    tup = eng.fetch_tuples(G * n * n)
    st = eng.stats()
    eng.close()
    assert len(games) == G and st["games_finished"] == G and st["tuples"] == sum(len(g["moves"]) for g in games) == tup["state"].shape[0]
    want = {}
    for g in games:
        for k, (turn, board, cell) in enumerate(pkg.replay_moves(n, g["moves"])):
            want.setdefault(board.tobytes(), set()).add((turn, g["winner"]))
    for i in range(0, tup["state"].shape[0], 7):
        key = tup["state"][i].tobytes()
        assert (int(tup["turn"][i]), int(tup["z"][i])) in want[key]
    assert tup["pi"].max() <= 4 * S * n        # child visits accumulate over reused subtrees, bounded by the sims played
    ws = np.array([g["winner"] for g in games])
    assert set(np.unique(ws)) <= {-1, 0, 1}


def test_net_batch_invariance_full_batch():
    """the resident net at a 32768-board pass (config 2's leaf batch): every board's output is independent of the batch
    it travels in (the conv kernel owns one board per workgroup), and the dihedral average of a constant policy is
    that constant (recover_way8 of eight equal images)."""
    need_gpu()
    import __graft_entry__ as ge
    n, C, blocks, G = 19, 256, 2, 4096
    blob = ge.make_random_blob(n, C, blocks, np.random.default_rng(3))
    eng = pkg.Engine(board_size=n, num_games=G, num_simulation=400, seed=4, evaluator=pkg.EVAL_NET, net_channels=C, net_blocks=blocks)
    eng.load_weights(blob)
    eng.step(1)                       # all 4096 leaves are the empty board: 32768 images in one pass
    r0 = eng.root(0)
    for g in (1, 777, 4095):
        r = eng.root(g)
        assert (r["prior"] == r0["prior"]).all() and r["root_value"] == r0["root_value"]
    pol, val = pkg.net_forward(n, C, blocks, blob, np.zeros((1, n * n), np.int8), -1)
    # the empty board is invariant under the 8 images, so the recovered prior is the dihedral average of ONE policy
    p = pol[0].reshape(n, n).astype(np.float64)
    avg = sum(np.rot90(q, k) for q in (p, np.fliplr(p)) for k in range(4)) / 8
    assert np.abs(r0["prior"].reshape(n, n) - avg).max() < 2e-7
    assert abs(r0["root_value"] - val[0]) < 1e-6
    eng.close()


def test_config1_full_game_19x19_400_sims():
    """BASELINE config 1 (1 game, 19x19, num_simulation = 400, RandomEvaluator, seed 0): the whole game -- every one of
    its ~200 stones chosen after 400 searches with tree reuse -- is bit-identical on the GPU engine and the CPU oracle."""
    need_gpu()
    import time
    n, S, seed = 19, 400, 0
    t0 = time.time()
    eng = pkg.Engine(board_size=n, num_games=1, num_simulation=S, seed=seed, evaluator=pkg.EVAL_RANDOM)
    eng.run()
    st = eng.stats()
    game = eng.fetch_games()[0]
    eng.close()
    t_gpu = time.time() - t0
    t0 = time.time()
    ref = o.self_play(n, num=1, threads=1, num_simulation=S, seed=seed)[0]
    t_cpu = time.time() - t0
    mv = [r * n + c for _, _, (r, c) in ref["path"]]
    assert list(game["moves"]) == mv and game["winner"] == ref["winner"]
    assert st["overflow_games"] == 0 and st["simulations"] == S * len(mv)
    print(f"config1: winner {ref['winner']}, {len(mv)} stones, oracle {t_cpu:.1f} s, engine (1 game) {t_gpu:.1f} s, "
          f"max records {st['max_records_used']}, max depth {st['max_depth']}")


def test_config2_exact():
    """BASELINE config 2 EXACTLY as bench.py times it: Engine(n = 19, G = 4096, S = 400, EVAL_NET, C = 256, stem + 3
    residual blocks, default precision = FP32 Winograd F(6,3), conv_wino6.hip), 3 rounds of 32 768 dihedral images each.  Every
    root has visit R and children R - 1; six sampled games -- first / last leaves of the batch and the seams of the Winograd
    scratch's 64-tile blocks -- equal the seeded oracle driven by its own FP32 net: visits exactly, priors / q / value to 1e-5."""
    need_gpu()
    import __graft_entry__ as ge
    n, G, S, C, blocks, R, seed = 19, 4096, 400, 256, 4, 3, 2024
    blob = ge.make_random_blob(n, C, blocks, np.random.default_rng(12))
    eng = pkg.Engine(board_size=n, num_games=G, num_simulation=S, seed=seed, evaluator=pkg.EVAL_NET, net_channels=C, net_blocks=blocks)
    eng.load_weights(blob)
    eng.step(R)
    st = eng.stats()
    assert st["simulations"] == G * R and st["overflow_games"] == 0 and st["live_games"] == G
    for g in list(range(0, G, 511)) + [G - 1]:
        r = eng.root(g)
        assert r["root_visit"] == R and r["visits"].sum() == R - 1 and r["stones"] == 0 and r["nsearch"] == R
    net = o.Net(n, C, blocks, blob)
    fn, ctx = net.eval_fn()
    empty = np.zeros((n, n), np.int8)
    for g in (0, 1, 799, 2048, 3333, G - 1):
        az = o.AlphaZero(n, num_simulation=S, eval_kind=o.EVAL_CUSTOM, eval_fn=fn, ctx=ctx, seed=seed, game_uid=g)
        s = o.sim_from_game(o.new_game(n))
        az.init(s)
        for _ in range(R):
            az.search(s)
        vis, q, p = az.children(empty)
        r = eng.root(g)
        assert (r["visits"] == vis).all(), (g, np.flatnonzero(r["visits"]), np.flatnonzero(vis))
        assert np.abs(r["q"] - q).max() <= 1e-5 and np.abs(r["prior"] - p).max() <= 1e-5
        assert abs(r["root_value"] - az.node_info(empty).value) <= 1e-5
        az.close()
    net.close()
    eng.close()


def test_config2_engine_on_distinct_midgame_boards():
    """test_config2_exact's leaves are the empty board and one-stone boards, whose eight dihedral images nearly
    coincide.  Here the SAME engine (n = 19, G = 4096, S = 400, EVAL_NET, C = 256, stem + 3 residual blocks, FP32 Winograd F(6,3)) first
    receives a different random position in every game -- 1 to 40 stones per game through c6_engine_apply_moves (Game::set's 1-2-2 turn
    order, so the side to move differs from game to game and both colours sit in one net pass) -- and then searches 3 rounds: every leaf of
    all 3 x 32 768-image passes is a distinct mid-game board.  Invariants on every 511th game; eight sampled games (first / last leaves,
    the 64-tile block seams, few and many stones) against the seeded oracle and its own FP32 net: visits exactly, priors / q / value to 1e-5."""
    need_gpu()
    import __graft_entry__ as ge
    n, G, S, C, blocks, R, seed, kmax = 19, 4096, 400, 256, 4, 3, 2025, 40
    cells = n * n
    rng = np.random.default_rng(77)
    nst = rng.integers(1, kmax + 1, G)
    nst[:4] = (1, 2, 39, 40)
    seq = np.stack([rng.permutation(cells)[:kmax] for _ in range(G)]).astype(np.int32)
    # no position may already hold a six (the game would be over): replay on the host and cut a game short of its winning stone
    for g in range(G):
        for k, (turn, board, cell) in enumerate(pkg.replay_moves(n, seq[g, :nst[g]])):
            b = board.copy().reshape(-1)
            b[cell] = turn
            if o.search(b.reshape(n, n)) != 0:
                nst[g] = k
                break
    assert nst.min() >= 1
    blob = ge.make_random_blob(n, C, blocks, np.random.default_rng(12))
    eng = pkg.Engine(board_size=n, num_games=G, num_simulation=S, seed=seed, evaluator=pkg.EVAL_NET, net_channels=C, net_blocks=blocks)
    eng.load_weights(blob)
    for k in range(kmax):
        eng.apply_moves(np.where(k < nst, seq[:, k], -2).astype(np.int32))
    eng.step(R)
    st = eng.stats()
    assert st["simulations"] == G * R and st["overflow_games"] == 0 and st["live_games"] == G
    for g in list(range(0, G, 511)) + [G - 1]:
        r = eng.root(g)
        assert r["root_visit"] == R and r["visits"][r["visits"] >= 0].sum() == R - 1 and r["stones"] == nst[g] and r["nsearch"] == R
    net = o.Net(n, C, blocks, blob)
    fn, ctx = net.eval_fn()
    for g in (0, 1, 2, 3, 799, 2048, 3333, G - 1):
        az = o.AlphaZero(n, num_simulation=S, eval_kind=o.EVAL_CUSTOM, eval_fn=fn, ctx=ctx, seed=seed, game_uid=g)
        game = o.new_game(n)
        for cell in seq[g, :nst[g]]:
            o.game_set(game, int(cell) // n, int(cell) % n)
        s = o.sim_from_game(game)
        az.init(s)
        for _ in range(R):
            az.search(s)
        board = game.np_board()
        r = eng.root(g)
        assert (r["board"] == board).all() and r["turn"] == game.turn, g
        vis, q, p = az.children(board)
        empty = board.reshape(-1) == 0
        assert (r["visits"][empty] == vis[empty]).all(), (g, np.flatnonzero(r["visits"] > 0), np.flatnonzero(vis > 0))
        assert np.abs(r["q"][empty] - q[empty]).max() <= 1e-5 and np.abs(r["prior"][empty] - p[empty]).max() <= 1e-5, g
        assert abs(r["root_value"] - az.node_info(board).value) <= 1e-5, g
        az.close()
    net.close()
    eng.close()


def test_config2_engine_moves_by_itself_with_the_net():
    """Config 2 as bench.py runs it -- staggered phases: slot g starts its first move with g mod S simulations already counted, so every round about
    G / S games make their move -- 6 rounds with the resident net (4096 games, S = 400, C = 256, stem + 3 blocks).  The games whose phase is S - 1 ... S - 6
    have then played their OWN first stone (policy() on the root's child visits, K1, next()'s prune and re-rooting, all under EVAL_NET at the
    benchmark's batch) and searched on from the new root: their leaves are one- and two-stone boards reached by the engine itself.  Sampled slots of
    every such phase against the oracle doing the same thing -- S - k searches, policy, Game::set, prune, the remaining searches: the stone, then the new
    root's child visits exactly and q / priors / value to 1e-5."""
    need_gpu()
    import __graft_entry__ as ge
    n, G, S, C, blocks, R, seed = 19, 4096, 400, 256, 4, 6, 2026
    blob = ge.make_random_blob(n, C, blocks, np.random.default_rng(12))
    eng = pkg.Engine(board_size=n, num_games=G, num_simulation=S, seed=seed, evaluator=pkg.EVAL_NET, net_channels=C, net_blocks=blocks, stagger=True)
    eng.load_weights(blob)
    eng.step(R)
    st = eng.stats()
    moved = [g for g in range(G) if S - (g % S) <= R]
    assert st["simulations"] == G * R and st["overflow_games"] == 0 and st["live_games"] == G and st["stones"] == len(moved) == 6 * (G // S) + sum(g % S >= S - 6 for g in range(G - G % S, G))
    net = o.Net(n, C, blocks, blob)
    fn, ctx = net.eval_fn()
    for g in (394, 395, 396, 397, 398, 399, 799, 2397, 3999, 0, 1):
        k = g % S
        first = S - k                                            # searches before the slot's first move
        az = o.AlphaZero(n, num_simulation=S, eval_kind=o.EVAL_CUSTOM, eval_fn=fn, ctx=ctx, seed=seed, game_uid=g)
        game = o.new_game(n)
        s = o.sim_from_game(game)
        az.init(s)
        for _ in range(min(first, R)):
            az.search(s)
        r = eng.root(g)
        if first <= R:
            cell = az.policy(s)
            az.prune(s)
            o.game_set(game, cell // n, cell % n)
            s = o.sim_from_game(game)
            az.init(s)
            for _ in range(R - first):
                az.search(s)
            assert r["stones"] == 1 and r["board"].reshape(-1)[cell] == -1 and (r["board"] != 0).sum() == 1, (g, cell)
        else:
            assert r["stones"] == 0
        board = game.np_board()
        assert (r["board"] == board).all() and r["turn"] == game.turn and r["remain"] == game.num_remain, g
        vis, q, p = az.children(board)
        empty = board.reshape(-1) == 0
        if az.node_info(board).found and az.node_info(board).has_prob:
            assert (r["visits"][empty] == vis[empty]).all(), (g, np.flatnonzero(r["visits"] > 0), np.flatnonzero(vis > 0))
            assert np.abs(r["q"][empty] - q[empty]).max() <= 1e-5 and np.abs(r["prior"][empty] - p[empty]).max() <= 1e-5, g
            assert abs(r["root_value"] - az.node_info(board).value) <= 1e-5, g
        else:      # (the slot moved in the last round: its new root has not been searched yet)
            assert first == R and r["root_visit"] == 0, g
        az.close()
    net.close()
    eng.close()
