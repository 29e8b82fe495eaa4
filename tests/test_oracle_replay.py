"""Oracle replay buffer against a literal Python model of AlphaZero/cpp_weighted/ReplayBuffer.hpp:22-37 (ring semantics)
and of its sample() output shapes (:39-65)."""
import numpy as np

import c6oracle as o


def games(n, count, seed):
    return o.self_play(n, num=count, threads=1, num_simulation=2, seed=seed)


def test_ring_semantics_match_reference_model():
    n, max_size = 9, 100
    rb = o.Replay(n, max_size)
    model, tail, num_data = [None] * max_size, 0, 0          # ReplayBuffer.hpp:22-29 transcribed
    for g in games(n, 6, 1):
        rb.push_game(g["winner"], g["path"])
        num_data = min(num_data + len(g["path"]), max_size)
        for p in g["path"]:
            model[tail] = (g["winner"], p)
            tail = (tail + 1) % max_size
        assert rb.size() == (num_data, tail)
    assert num_data == max_size                               # wrapped at least once
    w, pl, b, pos = rb.sample(64, seed=3)
    for i in range(64):
        idx = o.philox([i, 0, 0, 5], [3, 0])[0] % num_data
        win, (turn, board, (row, col)) = model[idx]
        assert (w[i], pl[i], pos[i]) == (win, turn, row * n + col) and (b[i].reshape(n, n) == board).all()
    rb.clear()
    assert rb.size() == (0, 0)
    rb.close()
