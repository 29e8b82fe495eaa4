"""BASELINE configs the driver had not exercised: config 3 at FULL size (16 384 concurrent 19x19 games, 800 simulations per
move, stem + 7 residual blocks) and the N > 1 path with REAL engines (config 4's mechanism: games sharded by uid, tuples
pooled by an all-gather) as two fresh processes on one GPU."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import c6oracle as o
from gpu_common import need_gpu, pkg

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


CONFIG3_SAMPLES = (0, 1, 511, 2047, 4095, 4096, 5000, 8191, 8192, 12287, 12288, 16383)


def test_config3_full_size():
    """16 384 games x 19x19 x S = 800 x 8-block resident net (131 072 dihedral images per round in 4 chunks of 32 768, ~80 GB of
    node records): after R rounds every root has visit R and its children R - 1 (alphazero_policy/tests.rs:90-136 generalised),
    nothing overflowed; twelve sampled games equal the seeded oracle driven by its own FP32 net (visits exactly, priors / q to
    1e-5).  The samples are the first, last and middle leaves of the net's chunks: their eight images sit in the first and the
    last (partial) block of each of the Winograd scratch's tile regions and on the chunk boundaries."""
    need_gpu()
    import __graft_entry__ as ge
    n, G, S, C, blocks, R, seed = 19, 16384, 800, 256, 8, 3, 31
    blob = ge.make_random_blob(n, C, blocks, np.random.default_rng(8))
    eng = pkg.Engine(board_size=n, num_games=G, num_simulation=S, seed=seed, evaluator=pkg.EVAL_NET, net_channels=C, net_blocks=blocks)
    eng.load_weights(blob)
    eng.step(R)
    st = eng.stats()
    assert st["simulations"] == G * R and st["overflow_games"] == 0 and st["live_games"] == G
    assert st["max_records_used"] <= R
    for g in list(range(0, G, 1021)) + [G - 1]:
        r = eng.root(g)
        assert r["root_visit"] == R and r["visits"].sum() == R - 1 and r["stones"] == 0 and r["nsearch"] == R
    net = o.Net(n, C, blocks, blob)
    fn, ctx = net.eval_fn()
    empty = np.zeros((n, n), np.int8)
    for g in CONFIG3_SAMPLES:
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


def test_config3_geometry_mid_game():
    """The same geometry (16 384 games, 19x19, S = 800) with the uniform evaluator through two whole moves and all but the last
    search of the third, so that mid-game trees, next()'s prune, tree reuse and the hand-over of record pages between games run
    at that size: every game has placed 2 stones, simulations == G * rounds, nothing overflowed, and sampled games' roots equal
    the seeded oracle's at the same point (child visits, q and priors bit for bit)."""
    need_gpu()
    n, G, S, seed, moves = 19, 16384, 800, 32, 2
    rounds = moves * S + S - 1
    eng = pkg.Engine(board_size=n, num_games=G, num_simulation=S, seed=seed, evaluator=pkg.EVAL_UNIFORM)
    eng.step(rounds)
    st = eng.stats()
    assert st["simulations"] == G * rounds and st["overflow_games"] == 0 and st["abandoned_games"] == 0 and st["live_games"] == G
    assert st["stones"] == G * moves
    for g in (0, 4097, G - 1):
        az = o.AlphaZero(n, num_simulation=S, eval_kind=o.EVAL_UNIFORM, seed=seed, game_uid=g)
        game = o.new_game(n)
        for _ in range(moves):
            cell = az.next(game)
            o.game_set(game, cell // n, cell % n)
        s = o.sim_from_game(game)
        az.init(s)
        for _ in range(S - 1):
            az.search(s)
        r = eng.root(g)
        board = game.np_board()
        assert (r["board"] == board).all(), g
        vis, q, p = az.children(board)
        empty = board.reshape(-1) == 0
        assert (r["visits"][empty] == vis[empty]).all() and (r["q"][empty] == q[empty]).all() and (r["prior"][empty] == p[empty]).all(), g
        az.close()
    eng.close()


def test_config3_geometry_deep_trees_share_the_pool():
    """config 3's geometry (16 384 games, 19x19, S = 800) on the stress evaluator with every 100th game peaked:
    those games' trees grow deep and keep more than 4 x S records alive while the others stay small -- they take their pages from
    the shared pool at move boundaries (nothing is copied, no other game is touched).  Memory stays under 200 GB, no game is
    abandoned or left parked, and the deep games -- and a shallow one -- equal the oracle's unbounded map (root children bit for bit)."""
    need_gpu()
    n, G, S, seed, eps, every, moves = 19, 16384, 800, 41, 0.001, 100, 6
    eng = pkg.Engine(board_size=n, num_games=G, num_simulation=S, seed=seed, epsilon=eps, evaluator=pkg.EVAL_STRESS, stress_every=every)
    for _ in range(moves):
        eng.step(S)
        eng.maintain()
    eng.step(S - 1)
    st = eng.stats()
    rounds = moves * S + S - 1
    assert st["simulations"] == G * rounds and st["stones"] == G * moves, st
    assert st["abandoned_games"] == 0 and st["parked_games"] == 0 and st["overflow_games"] == 0 and st["live_games"] == G
    assert st["max_records_used"] > 4 * S and st["max_pages_held"] * 64 > 4 * S, st
    assert st["pool_bytes"] < 200e9, st

    def oracle_stress(ctx, turn, nn, ln, boards, values, policies):
        b = np.ctypeslib.as_array(boards, shape=(ln, nn * nn))
        v, p = pkg.stress_eval(nn, turn, b)
        np.ctypeslib.as_array(values, shape=(ln,))[:] = v
        np.ctypeslib.as_array(policies, shape=(ln, nn * nn))[:] = p
        return 0

    for g in (0, 100, 16300, 7):                                 # three peaked games, one uniform
        peaked = g % every == 0
        az = o.AlphaZero(n, num_simulation=S, epsilon=eps, eval_kind=o.EVAL_CUSTOM if peaked else o.EVAL_UNIFORM,
                         eval_fn=oracle_stress if peaked else None, seed=seed, game_uid=g)
        game = o.new_game(n)
        for _ in range(moves):
            cell = az.next(game)
            o.game_set(game, cell // n, cell % n)
        s = o.sim_from_game(game)
        az.init(s)
        for _ in range(S - 1):
            az.search(s)
        r = eng.root(g)
        board = game.np_board()
        assert (r["board"] == board).all(), g
        vis, q, p = az.children(board)
        empty = board.reshape(-1) == 0
        assert (r["visits"][empty] == vis[empty]).all() and (r["q"][empty] == q[empty]).all() and (r["prior"][empty] == p[empty]).all(), g
        if peaked:
            assert r["records_used"] > 4 * S, (g, r["records_used"])
        az.close()
    eng.close()


def test_config5_engine_full_size():
    """BASELINE config 5 as an ENGINE at full size: 32 768 concurrent 9x9 games, S = 100, uniform evaluator, recycling slots,
    for about two game lengths of rounds through c6_search_kernel<2, 2> (the kernel the config-5 bench times).  Nothing
    overflowed or was abandoned, simulations == G * rounds, every finished game replays legally on K1 with K1's winner and
    length (a game ends exactly at its first six / on the full board), and sampled games -- first-generation and recycled
    uids -- equal the seeded oracle stone for stone."""
    need_gpu()
    n, G, S, seed, rounds = 9, 32768, 100, 77, 13200
    eng = pkg.Engine(board_size=n, num_games=G, num_simulation=S, seed=seed, evaluator=pkg.EVAL_UNIFORM, recycle=True)
    eng.step(rounds)
    st = eng.stats()
    games = eng.fetch_games()
    eng.close()
    assert st["simulations"] == G * rounds and st["overflow_games"] == 0 and st["abandoned_games"] == 0 and st["live_games"] == G
    assert len(games) == st["games_finished"] and len(games) > G
    uids = np.array([g["uid"] for g in games])
    assert len(set(uids.tolist())) == len(games) and (uids % G).size       # every uid finishes once
    # legality + winner: all finished games replayed in lock step through K1
    F = len(games)
    Fp = (F + 3) & ~3
    lens = np.array([len(g["moves"]) for g in games] + [0] * (Fp - F))
    mv = np.full((Fp, n * n), -2, np.int32)
    for i, g in enumerate(games):
        mv[i, :len(g["moves"])] = g["moves"]
    soa = np.zeros((n * n, Fp), np.int8)
    turn = np.full(Fp, -1, np.int8)
    remain = np.ones(Fp, np.int8)
    won_at = np.full(Fp, -1, np.int32)
    winner = np.zeros(Fp, np.int8)
    for k in range(n * n):
        pos = np.ascontiguousarray(mv[:, k])
        if (pos < 0).all():
            break
        soa, turn, remain, win, err = pkg.board_step(n, soa, turn, remain, pos)
        live = pos >= 0
        assert (err[live] == 0).all(), k
        first = live & (win != 0) & (won_at < 0)
        won_at[first] = k
        winner[first] = win[first]
    want_w = np.array([g["winner"] for g in games] + [0] * (Fp - F))
    assert (winner == want_w).all()
    decided = want_w != 0
    assert (won_at[decided] == lens[decided] - 1).all()                      # the game stopped at its first six
    assert (lens[~decided][:F - decided.sum()] <= n * n).all()
    draws = (~decided)[:F]
    assert (lens[:F][draws] == n * n).all()                                  # no winner: the board is full (policy() had no move)
    # stone for stone against the oracle: slots' first games and recycled ones
    by_uid = {g["uid"]: g for g in games}
    first_gen = [u for u in (0, 1, 12345, G - 1) if u in by_uid]
    recycled = sorted(u for u in by_uid if u >= G)[:4]
    assert len(first_gen) >= 3 and len(recycled) == 4
    for u in first_gen + recycled:
        ref = o.self_play(n, num=1, threads=1, num_simulation=S, seed=seed, eval_kind=o.EVAL_UNIFORM, uid_base=u)[0]
        want = [r * n + c for _, _, (r, c) in ref["path"]]
        assert list(by_uid[u]["moves"]) == want and by_uid[u]["winner"] == ref["winner"], u


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,per_rank", [(2, 6), (4, 8)], ids=["world2", "world4-32uids"])
def test_multi_process_engine_shard_on_one_gpu(tmp_path, world, per_rank):
    """The N > 1 path with real engines: `world` fresh child processes (gloo, all on GPU 0) each play their shard of the games
    (shard.uid_plan) on their own engine and pool the training tuples with shard.TuplePool; every rank must end up with the
    same pooled set, and that set must equal the single-process run of all the games -- a concatenation in any order, as
    AsyncAgent::run collects results (agent/async_agent/mod.rs:100-137).  world 4 x 8 games = the 32 uids VERDICT r3 asked for
    as 8 x 4: the GPU box admits at most 6 processes on its card (this one included), so 4 ranks is what a test may spawn; the
    8-rank sharding plan and all-gather run over gloo on the CPU (tests/test_dist_gloo.py) and 4 ranks x bench.py in
    tools/share_rehearsal.sh."""
    need_gpu()
    n, S, seed = 9, 12, 77
    port = _free_port()
    outs = [str(tmp_path / f"rank{r}.npz") for r in range(world)]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "shard_worker.py"), str(r), str(world), str(port), str(n),
                               str(per_rank), str(S), str(seed), outs[r], "0", "gloo"], env=env) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    # single-process run of the same uids
    G = per_rank * world
    eng = pkg.Engine(board_size=n, num_games=G, num_simulation=S, seed=seed, evaluator=pkg.EVAL_RANDOM)
    eng.enable_tuples(G * n * n)
    eng.run()
    games = eng.fetch_games()
    stride = pkg.lib().c6_tuple_stride(n)
    buf = np.zeros((G * n * n, stride), np.uint8)
    import ctypes as C
    cnt = C.c_size_t()
    assert pkg.lib().c6_engine_fetch_tuples(eng.h, G * n * n, C.byref(cnt), buf.ctypes.data_as(C.c_void_p)) == 0
    eng.close()
    want = sorted(bytes(r) for r in buf[: cnt.value])
    assert len(want) == sum(len(g["moves"]) for g in games)
    got = [np.load(f) for f in outs]
    for r in range(world):
        assert int(got[r]["overflow"]) == 0 and int(got[r]["finished"]) == per_rank and int(got[r]["dropped"]) == 0
        assert list(got[r]["uids"]) == list(range(r * per_rank, (r + 1) * per_rank))       # shard.uid_plan: a partition of the uids
        assert sorted(bytes(x) for x in got[r]["records"]) == want                          # every rank holds the whole pooled set
    # the pooled set replays to the oracle's games too (engine == oracle is tested elsewhere; this ties the shard to it)
    ref = o.self_play(n, num=G, threads=min(G, os.cpu_count() or 1), num_simulation=S, seed=seed)
    assert sorted(len(g_["path"]) for g_ in ref) == sorted(len(g["moves"]) for g in games)


def _spawn_ranks(tmp_path, world, n, per_rank, S, seed, backend, mode, devices, evaluator="random"):
    port = _free_port()
    outs = [str(tmp_path / f"{mode}_{evaluator}_rank{r}.npz") for r in range(world)]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "shard_worker.py"), str(r), str(world), str(port), str(n),
                               str(per_rank), str(S), str(seed), outs[r], str(devices[r]), backend, mode, evaluator], env=env) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=900) == 0
    return [np.load(f) for f in outs]


def _check_net_samples(got, n, S, seed):
    """the roots a net-driven rank stored (shard_worker.props) against the oracle driven by its own FP32 net: visits exactly,
    priors / q / value to 1e-5 (north_star's tolerance for the net)"""
    import __graft_entry__ as ge
    import shard_worker as sw
    blob = ge.make_random_blob(n, sw.NET_C, sw.NET_BLOCKS, np.random.default_rng(sw.NET_BLOB_SEED))
    net = o.Net(n, sw.NET_C, sw.NET_BLOCKS, blob)
    fn, ctx = net.eval_fn()
    empty = np.zeros((n, n), np.int8)
    for i, uid in enumerate(got["sample_uids"]):
        az = o.AlphaZero(n, num_simulation=S, eval_kind=o.EVAL_CUSTOM, eval_fn=fn, ctx=ctx, seed=seed, game_uid=int(uid))
        s = o.sim_from_game(o.new_game(n))
        az.init(s)
        for _ in range(sw.PROPS_ROUNDS):
            az.search(s)
        vis, q, p = az.children(empty)
        assert (got["sample_visits"][i] == vis).all(), (int(uid), np.flatnonzero(got["sample_visits"][i]), np.flatnonzero(vis))
        assert np.abs(got["sample_q"][i] - q).max() <= 1e-5 and np.abs(got["sample_prior"][i] - p).max() <= 1e-5, int(uid)
        assert abs(float(got["sample_value"][i]) - az.node_info(empty).value) <= 1e-5, int(uid)
        az.close()
    net.close()


def _config4_checks(tmp_path, backend, devices, per_rank_props):
    world = 2
    # (a) whole games: every rank ends with the same pooled set = the single-process run of all the uids
    n, per_rank, S, seed = 9, 6, 12, 78
    got = _spawn_ranks(tmp_path, world, n, per_rank, S, seed, backend, "play", devices)
    G = per_rank * world
    eng = pkg.Engine(board_size=n, num_games=G, num_simulation=S, seed=seed, evaluator=pkg.EVAL_RANDOM)
    eng.enable_tuples(G * n * n)
    eng.run()
    games = eng.fetch_games()
    stride = pkg.lib().c6_tuple_stride(n)
    buf = np.zeros((G * n * n, stride), np.uint8)
    import ctypes as C
    cnt = C.c_size_t()
    assert pkg.lib().c6_engine_fetch_tuples(eng.h, G * n * n, C.byref(cnt), buf.ctypes.data_as(C.c_void_p)) == 0
    eng.close()
    want = sorted(bytes(r) for r in buf[: cnt.value])
    assert len(want) == sum(len(g["moves"]) for g in games)
    for r in range(world):
        assert int(got[r]["overflow"]) == 0 and int(got[r]["finished"]) == per_rank
        assert list(got[r]["uids"]) == list(range(r * per_rank, (r + 1) * per_rank))
        assert sorted(bytes(x) for x in got[r]["records"]) == want
    # (b) config 4's shape: many 19x19 games per rank for a few rounds, then one game per rank finished from outside
    got = _spawn_ranks(tmp_path, world, 19, per_rank_props, 400, seed, backend, "props", devices)
    for r in range(world):
        assert int(got[r]["ok"]) == 1, r                      # root visit == R, sum(child visits) == R - 1, nothing overflowed
        assert int(got[r]["gathers"]) >= 1 and int(got[r]["dropped"]) == 0
        assert int(got[r]["finished"]) == 1 and int(got[r]["winner"]) == 1        # the scripted game: White's six
        assert int(got[r]["pooled"]) == 11 * world and int(got[r]["total"]) == 11 * world
    assert sorted(bytes(x) for x in got[0]["records"]) == sorted(bytes(x) for x in got[1]["records"])


def test_config4_two_gpu_rccl(tmp_path):
    """BASELINE config 4's mechanism on real RCCL: 2 fresh child processes, ONE GPU EACH, backend nccl -- the device-tensor branch
    of shard.TuplePool (all-gather of the counts, then of max(count) records, on a side stream).  Runs whenever the box has
    at least two GPUs; the one-GPU box of this pool skips it (and runs the same protocol over gloo, below)."""
    need_gpu()
    if pkg.device_count() < 2:
        pytest.skip("needs 2 GPUs: RCCL wants one device per rank")
    _config4_checks(tmp_path, "nccl", (0, 1), 8192)


def test_config4_per_gpu_workload(tmp_path):
    """BASELINE config 4's PER-GPU workload assembled in one process (missing 4): Engine(n = 19, G = 8192, S = 400,
    EVAL_NET, C = 256, stem + 3 residual blocks) + shard.TuplePool on the nccl backend.  The rank builds a ONE-rank RCCL group and the pool
    issues its all-gathers through it (collective=True), so the device-tensor branch -- counts, then max(count) records, on the side stream,
    ordered against the engine's stream by events -- executes exactly as a rank of eight would execute it, as far as one GPU can.
    6 rounds of 65 536 dihedral images; root invariants on samples; the scripted 11-stone game finished from outside so that the side
    stream carries a payload; four sampled games (first / last leaves of the two net passes) against the oracle and its own FP32 net."""
    need_gpu()
    n, G, S, seed = 19, 8192, 400, 79
    got = _spawn_ranks(tmp_path, 1, n, G, S, seed, "nccl", "props", (0,), evaluator="net")[0]
    assert int(got["ok"]) == 1 and int(got["collective"]) == 1
    assert int(got["gathers"]) >= 1 and int(got["dropped"]) == 0 and int(got["exchanges"]) >= 3
    assert int(got["finished"]) == 1 and int(got["winner"]) == 1
    assert int(got["pooled"]) == 11 and int(got["total"]) == 11
    # the pooled records ARE the scripted game: 11 states, each the replay of the script's prefix
    tup = pkg.decode_tuples(got["records"], n)
    script = [(0, 0), (5, 0), (5, 1), (0, 1), (0, 2), (5, 2), (5, 3), (0, 3), (0, 4), (5, 4), (5, 5)]
    want = list(pkg.replay_moves(n, [r * n + c for r, c in script]))
    order = np.argsort((tup["state"].reshape(len(want), -1) != 0).sum(axis=1), kind="stable")
    for k, i in enumerate(order):
        turn, board, cell = want[k]
        assert (tup["state"][i] == board).all() and int(tup["turn"][i]) == turn and int(tup["pos"][i]) == cell and int(tup["z"][i]) == 1
    assert list(got["sample_slots"]) == [0, G // 2 - 1, G // 2, G - 1]
    _check_net_samples(got, n, S, seed)


def test_config4_two_ranks_one_gpu_gloo_with_the_net(tmp_path):
    """the two-rank stand-in with the RESIDENT NET on both ranks (round 5's stand-in ran the random evaluator only): 2 x 2048 games of
    config 4's shape on GPU 0, tuples pooled over gloo, both ranks' sampled roots against the oracle -- rank 1's uids start at 2048
    (shard.uid_plan), so its games are not rank 0's"""
    need_gpu()
    n, per_rank, S, seed = 19, 2048, 400, 80
    got = _spawn_ranks(tmp_path, 2, n, per_rank, S, seed, "gloo", "props", (0, 0), evaluator="net")
    for r in range(2):
        assert int(got[r]["ok"]) == 1 and int(got[r]["dropped"]) == 0 and int(got[r]["gathers"]) >= 1
        assert int(got[r]["finished"]) == 1 and int(got[r]["winner"]) == 1 and int(got[r]["pooled"]) == 22 and int(got[r]["total"]) == 22
        assert list(got[r]["sample_uids"]) == [r * per_rank + g for g in (0, per_rank // 2 - 1, per_rank // 2, per_rank - 1)]
    assert sorted(bytes(x) for x in got[0]["records"]) == sorted(bytes(x) for x in got[1]["records"])
    _check_net_samples(got[0], n, S, seed)
    _check_net_samples({k: got[1][k][-1:] for k in ("sample_uids", "sample_visits", "sample_q", "sample_prior", "sample_value")}, n, S, seed)


def test_config4_two_ranks_one_gpu_gloo(tmp_path):
    """the same checks with both ranks on GPU 0 (gloo carries the all-gathers; RCCL refuses two ranks on one device), at
    2 x 2048 games of config 4's shape so that both engines fit beside the other tests"""
    need_gpu()
    _config4_checks(tmp_path, "gloo", (0, 0), 2048)
