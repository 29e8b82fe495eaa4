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


def test_config3_full_size():
    """16 384 games x 19x19 x S = 800 x 8-block resident net (131 072 dihedral images per round, ~80 GB of node records):
    after R rounds every root has visit R and its children R - 1 (alphazero_policy/tests.rs:90-136 generalised), nothing
    overflowed; three sampled games equal the seeded oracle driven by its own FP32 net (visits exactly, priors / q to 1e-5)."""
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
    for g in (0, 5000, G - 1):
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


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_process_engine_shard_on_one_gpu(tmp_path):
    """The N > 1 path with real engines: 2 fresh child processes (gloo, both on GPU 0) each play their shard of the games
    (shard.uid_plan) on their own engine and pool the training tuples with shard.TuplePool; every rank must end up with the
    same pooled set, and that set must equal the single-process run of all the games -- a concatenation in any order, as
    AsyncAgent::run collects results (agent/async_agent/mod.rs:100-137)."""
    need_gpu()
    n, per_rank, S, seed, world = 9, 6, 12, 77, 2
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
        assert int(got[r]["overflow"]) == 0 and int(got[r]["finished"]) == per_rank
        assert list(got[r]["uids"]) == list(range(r * per_rank, (r + 1) * per_rank))       # shard.uid_plan: a partition of the uids
        assert sorted(bytes(x) for x in got[r]["records"]) == want                          # every rank holds the whole pooled set
    # the pooled set replays to the oracle's games too (engine == oracle is tested elsewhere; this ties the shard to it)
    ref = o.self_play(n, num=G, threads=min(G, os.cpu_count() or 1), num_simulation=S, seed=seed)
    assert sorted(len(g_["path"]) for g_ in ref) == sorted(len(g["moves"]) for g in games)
