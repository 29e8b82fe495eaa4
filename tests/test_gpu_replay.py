"""Device-resident replay buffer against the oracle's restatement of
AlphaZero/cpp_weighted/ReplayBuffer.hpp: ring semantics and sampled batches bit-identical."""
import ctypes as C

import numpy as np
import pytest

import c6oracle as o
from gpu_common import need_gpu, pkg

pytestmark = pytest.mark.gpu


def test_push_and_sample_match_oracle(n):
    need_gpu()
    max_size = 150
    games = o.self_play(n, num=5, threads=1, num_simulation=2, seed=n, want_pi=True)
    rb, ref = pkg.ReplayBuffer(n, max_size), o.Replay(n, max_size)
    for g in games:                                           # wraps the ring (push_game, ReplayBuffer.hpp:22-29)
        rb.push_game((g["winner"], g["path"]), pi=g["pi"])
        ref.push_game(g["winner"], g["path"])
        assert rb.size() == ref.size()
    for draw in range(3):
        w, p, b, q, pi = rb.sample(97, seed=11, draw_id=draw, want_pi=True)
        rw, rp, rbd, rq = ref.sample(97, seed=11, draw_id=draw)
        assert (w == rw).all() and (p == rp).all() and (b == rbd).all() and (q == rq).all()
        assert ((b != 0).sum(axis=1) <= n * n).all() and (pi >= 0).all()
    rb.clear()
    assert rb.size() == (0, 0)
    rb.close()
    ref.close()


def test_engine_tuples_feed_the_buffer_on_device():
    """self-play -> tuple buffer -> replay ring without a host round trip (c6_engine_push_tuples); every sampled row is a
    position of one of the finished games with that game's winner and the stone actually played there"""
    need_gpu()
    n, G, S = 9, 64, 4
    eng = pkg.Engine(board_size=n, num_games=G, num_simulation=S, seed=5, evaluator=pkg.EVAL_UNIFORM)
    eng.enable_tuples(G * n * n)
    eng.run()
    games = eng.fetch_games()
    total = sum(len(g["moves"]) for g in games)
    assert eng.stats()["tuples"] == total
    rb = pkg.ReplayBuffer(n, 4096)
    eng.push_tuples(rb)
    assert rb.size() == (min(total, 4096), total % 4096)
    assert eng.fetch_tuples(16)["state"].shape[0] == 0            # the engine's counter was reset by the push
    truth = {}
    for g in games:
        for turn, board, cell in pkg.replay_moves(n, g["moves"]):
            truth.setdefault(board.tobytes(), set()).add((turn, g["winner"], cell))
    w, p, b, q = rb.sample(256, seed=1)
    for i in range(256):
        key = b[i].astype(np.int8).reshape(n, n).tobytes()
        assert (int(p[i]), int(w[i]), int(q[i])) in truth[key]
    eng.close()
    rb.close()
