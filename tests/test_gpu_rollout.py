"""DefaultPolicy on the GPU (include/connect6_amd.h part 5, csrc/rollout.hip) against oracle/c6_default.c: moves, trees
and whole games bit-identical on the same C6RNG keys; the reference's own checks (default_policy/tests.rs,
multi_policy/tests.rs) through the drop-in `cpp_play`."""
import ctypes as C

import numpy as np
import pytest

import c6oracle as o
from gpu_common import need_gpu, pkg

pytestmark = pytest.mark.gpu

B, Z, W = o.BLACK, o.NONE, o.WHITE


def start_positions(n, G, rng, max_stones):
    """G oracle games advanced by a random number of random legal stones (no winner yet)"""
    games = []
    for g in range(G):
        while True:
            gm = o.new_game(n)
            for _ in range(int(rng.integers(0, max_stones + 1))):
                idx = np.flatnonzero(gm.np_board().reshape(-1) == 0)
                c = int(rng.choice(idx))
                o.game_set(gm, c // n, c % n)
            if o.search(gm.np_board()) == Z:
                break
        games.append(gm)
    return games


def boards_of(games):
    return np.stack([g.np_board().reshape(-1) for g in games]).astype(np.int8)


@pytest.mark.parametrize("n,G,iters,moves", [(9, 96, 12, 6), (15, 48, 20, 4), (19, 24, 50, 3)])
def test_next_matches_oracle_moves_and_trees(n, G, iters, moves):
    need_gpu()
    rng = np.random.default_rng(n)
    games = start_positions(n, G, rng, 20)
    dp = pkg.DefaultPolicy(n, G, num_iter=iters, seed=0x1234567887654321)
    refs = [o.DefaultPolicy(n, num_iter=iters, seed=0x1234567887654321, uid=g) for g in range(G)]
    live = np.ones(G, np.uint8)
    for mv in range(moves):
        pos = dp.next(boards_of(games), active=live)
        for g in range(G):
            if not live[g]:
                assert pos[g] == -1
                continue
            want = refs[g].next(games[g])
            assert pos[g] == want, (mv, g)
            # the whole root node: visits, Black's wins, children in push order with their counts
            assert dp.node(g, games[g].np_board()) == refs[g].node(games[g].np_board()), (mv, g)
            info = dp.info(g)
            assert info["nodes"] == refs[g].map_len() and info["searches"] == refs[g].search_count() and info["uid"] == g
            o.game_set(games[g], want // n, want % n)
            if o.search(games[g].np_board()) != Z:
                live[g] = 0
    dp.close()


@pytest.mark.parametrize("n,G,iters", [(9, 128, 8), (15, 16, 10)])
def test_whole_games_match_oracle(n, G, iters):
    """Agent::play with one DefaultPolicy on both sides (default_policy/tests.rs:130-146), G games side by side"""
    need_gpu()
    dp = pkg.DefaultPolicy(n, G, num_iter=iters, seed=77)
    games = [o.new_game(n) for _ in range(G)]
    live = np.ones(G, np.uint8)
    moves = [[] for _ in range(G)]
    winners = np.zeros(G, np.int8)
    while live.any():
        pos = dp.next(boards_of(games), active=live)
        for g in np.flatnonzero(live):
            if pos[g] < 0:
                live[g] = 0
                continue
            moves[g].append(int(pos[g]))
            o.game_set(games[g], pos[g] // n, pos[g] % n)
            w = o.search(games[g].np_board())
            if w != Z:
                winners[g], live[g] = w, 0
    for g in range(G):
        ref = o.DefaultPolicy(n, num_iter=iters, seed=77, uid=g)
        r = o.agent_play(n, lambda gm: (lambda c: None if c < 0 else (c // n, c % n))(ref.next(gm)))
        assert [p[0] * n + p[1] for _, _, p in r["path"]] == moves[g], g
        assert r["winner"] == winners[g]
        if r["winner"] != Z:
            assert r["path"][-1][0] == r["winner"]          # last.turn == winner
        assert dp.info(g)["searches"] == ref.search_count()
    dp.close()


def test_reset_starts_a_fresh_policy_on_the_next_uid():
    need_gpu()
    n, G, iters = 9, 8, 10
    dp = pkg.DefaultPolicy(n, G, num_iter=iters, seed=5)
    games = start_positions(n, G, np.random.default_rng(1), 10)
    first = dp.next(boards_of(games))
    mask = np.zeros(G, np.uint8)
    mask[[1, 4]] = 1
    dp.reset(mask)
    again = dp.next(boards_of(games))
    for g in range(G):
        if mask[g]:
            ref = o.DefaultPolicy(n, num_iter=iters, seed=5, uid=g + G)
            assert again[g] == ref.next(games[g])
            info = dp.info(g)
            assert (info["nodes"], info["searches"], info["uid"]) == (ref.map_len(), iters, g + G)
        else:                                                 # the map lives on: 2 x num_iter searches on the same root
            ref = o.DefaultPolicy(n, num_iter=iters, seed=5, uid=g)
            assert first[g] == ref.next(games[g]) and again[g] == ref.next(games[g])
            assert dp.info(g)["searches"] == 2 * iters
    dp.close()


def test_finished_board_and_inactive_slots():
    need_gpu()
    n = 9
    g = o.new_game(n)
    for r, c in [(0, 0), (8, 0), (8, 1), (0, 1), (0, 2), (8, 3), (8, 4), (0, 3), (0, 4), (8, 6), (8, 7), (0, 5)]:
        o.game_set(g, r, c)
    assert o.search(g.np_board()) == B
    dp = pkg.DefaultPolicy(n, 2, num_iter=5, seed=9)
    boards = np.stack([g.np_board().reshape(-1), np.zeros(n * n, np.int8)])
    pos = dp.next(boards, active=np.array([1, 0], np.uint8))
    ref = o.DefaultPolicy(n, num_iter=5, seed=9, uid=0)
    assert pos[0] == ref.next(g) and pos[1] == -1             # mod.rs:224-226: nothing is searched, policy picks at random
    assert dp.node(0, g.np_board()) == (0, 0, []) and dp.info(0)["nodes"] == 1 and dp.info(1)["searches"] == 0
    full = np.ones(n * n, np.int8)
    full[::2] = -1
    assert dp.next(np.stack([full, full]))[0] == -1           # no empty cell: -1 (the reference would panic)
    dp.close()


def test_full_tree_is_an_error():
    need_gpu()
    n = 9
    dp = pkg.DefaultPolicy(n, 4, num_iter=10, seed=1, max_nodes=16)
    boards = np.zeros((4, n * n), np.int8)
    dp.next(boards)
    with pytest.raises(pkg.C6Error, match="is full"):
        dp.next(boards)
    dp.reset()
    dp.next(boards)                                           # usable again after DefaultPolicy::new
    dp.close()
    with pytest.raises(pkg.C6Error):
        pkg.DefaultPolicy(10, 1)


def test_self_play_through_cpp_play():
    """default_policy/tests.rs:130-146 over the drop-in boundary: Agent::play(DefaultPolicy::with_num_iter(10)), N = 15"""
    need_gpu()
    dp = pkg.DefaultPolicy(pkg.REF_N, 1, num_iter=10, seed=3)
    (winner, path), = pkg.play(dp)
    assert len(path) > 0 and dp.info()["searches"] == 10 * len(path)
    if winner != Z:
        assert path[-1][0] == winner
    final = np.array(path[-1][1], np.int8).reshape(pkg.REF_N, pkg.REF_N).copy()
    final[path[-1][2]] = path[-1][0]
    assert o.search(final) == winner
    dp.close()


def test_multi_policy_through_cpp_play():
    """multi_policy/tests.rs:8-27: Black = RandomPolicy, White = DefaultPolicy::with_num_iter(10)"""
    need_gpu()
    dp, rp = pkg.DefaultPolicy(pkg.REF_N, 1, num_iter=10, seed=4), pkg.RandomPolicy(seed=8)

    def multi(board):                                         # multi_policy/mod.rs:61-71; the turn from the stone count
        s = int(np.count_nonzero(board))
        white = s > 0 and ((s - 1) // 2) % 2 == 0
        return dp(board) if white else rp(board)

    (winner, path), = pkg.play(multi)
    if winner != Z:
        assert path[-1][0] == winner
    assert all(t == (W if (k > 0 and ((k - 1) // 2) % 2 == 0) else B) for k, (t, _, _) in enumerate(path))
    dp.close()


def test_full_size_properties():
    """BASELINE-size sanity where the oracle would take minutes: N = 19, 4096 policies, num_iter = 50 (DefaultPolicy::new)"""
    need_gpu()
    n, G = 19, 4096
    dp = pkg.DefaultPolicy(n, G, seed=2)
    boards = np.zeros((G, n * n), np.int8)
    boards[:, 180] = -1                                       # Black opened in the centre, White to move
    pos = dp.next(boards)
    assert ((pos >= 0) & (pos < n * n) & (pos != 180)).all()
    assert len(np.unique(pos)) > 100                          # keyed per slot
    for g in (0, 1, 4095):
        visit, bw, kids = dp.node(g, boards[g])
        info = dp.info(g)
        assert info["searches"] == 50 and 1 <= len(kids) <= info["edges"] <= 50 and info["nodes"] <= 51
        assert 0 <= bw <= visit <= 50
    again = pkg.DefaultPolicy(n, G, seed=2)
    assert (again.next(boards) == pos).all()                  # reproducible
    dp.close()
    again.close()


def test_playouts_match_a_host_replay_of_the_stream():
    need_gpu()
    n, G, iters = 15, 256, 3
    seed = 0xABCDEF0012345678
    rng = np.random.default_rng(0)
    games = start_positions(n, G, rng, 30)
    boards = boards_of(games)
    dp = pkg.DefaultPolicy(n, G, num_iter=0, seed=seed)
    ms, steps, win = dp.bench_playouts(boards, iters=iters)
    assert ms > 0 and steps > G and set(np.unique(win)) <= {-1, 0, 1}
    for g in range(0, G, 37):                                 # the last launch used search = iters - 1, uid = g
        gm = games[g]
        sim = o.sim_from_game(gm)
        d = 0
        while o.search(sim.np_board()) == Z and sim.npossible:
            w = o.philox([d >> 2, 0, iters - 1, 6], [seed & 0xFFFFFFFF, ((seed >> 32) + g) & 0xFFFFFFFF])[d & 3]
            cell = sim.possible_list()[(int(w) * sim.npossible) >> 32]
            o.lib().c6o_sim_simulate_in(sim, cell // n, cell % n)
            d += 1
        assert o.search(sim.np_board()) == win[g], g
    dp.close()
