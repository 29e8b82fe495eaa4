"""K-8: Connect6/libconnect6/src/policy/alphazero_policy/tests.rs:6-204 on the oracle, plus the
reference quirks of SURVEY section 0 (transposition merge, one-level backup, prune)."""
import ctypes as C

import numpy as np

import c6oracle as o

B, Z, W = o.BLACK, o.NONE, o.WHITE


def fresh(n, **kw):
    g = o.new_game(n)
    s = o.sim_from_game(g)
    az = o.AlphaZero(n, eval_kind=o.EVAL_RANDOM, seed=7, **kw)
    az.init(s)
    return g, s, az


def run_search_by_hand(az, s):
    L = o.lib()
    path = []
    while True:
        cell = az.select(s)
        if cell < 0:
            break
        assert cell in s.possible_list() and L.c6o_sim_validate(C.byref(s), cell // s.n, cell % s.n)
        path.append(cell)
        L.c6o_sim_simulate_in(C.byref(s), cell // s.n, cell % s.n)
    az.expand(s)
    az.update(s, path)
    return path


def test_select(n):  # tests.rs:6-40
    g, s, az = fresh(n)
    run_search_by_hand(az, s)
    cell = az.select(s)
    assert cell >= 0 and cell in s.possible_list()


def test_expand(n):  # tests.rs:42-88
    g, s, az = fresh(n)
    while az.select(s) >= 0:
        pass
    az.expand(s)
    empty = np.zeros((n, n), np.int8)
    root = az.node_info(empty)
    assert root.found and root.visit == 1 and root.value != 0 and root.q_value == 0 and root.n_prob == 0
    prob = az.node_prob(empty)
    assert prob.any() and root.num_player == 0 and root.n_next == n * n
    vis, q, nprob = az.children(empty)
    for i in range(n):
        for j in range(n):
            b = empty.copy()
            b[i, j] = B
            ch = az.node_info(b)
            assert ch.found and ch.visit == 0 and ch.value == 0 and ch.n_prob == prob[i, j]
            assert not ch.has_prob and ch.num_player == 1 and ch.n_next == 0
            assert nprob[i * n + j] == prob[i, j]


def two_searches(n):
    g = o.new_game(n)
    s0 = o.sim_from_game(g)
    az = o.AlphaZero(n, eval_kind=o.EVAL_RANDOM, seed=11)
    az.init(s0)
    for _ in range(2):
        s = o.sim_from_game(g)
        run_search_by_hand(az, s)
    return g, s0, az


def test_update(n):  # tests.rs:90-136
    g, s0, az = two_searches(n)
    empty = np.zeros((n, n), np.int8)
    root = az.node_info(empty)
    assert root.q_value != 0 and root.visit == 2 and root.n_next == n * n
    prob = az.node_prob(empty)
    hit = []
    for c in range(n * n):
        b = empty.copy().reshape(-1)
        b[c] = B
        ch = az.node_info(b.reshape(n, n))
        if ch.value != 0:
            hit.append((c, ch))
    assert len(hit) == 1
    c, ch = hit[0]
    assert ch.n_prob != 0 and np.float32(root.q_value) * np.float32(2) == np.float32(ch.value)
    assert prob.reshape(-1)[c] == ch.n_prob


def test_policy(n):  # tests.rs:138-183
    g, s0, az = two_searches(n)
    cell = az.policy(s0)
    assert cell >= 0 and cell in s0.possible_list()
    empty = np.zeros((n, n), np.int8)
    hit = [c for c in range(n * n) if az.node_info(_with(empty, c, B)).value != 0]
    assert hit == [cell]


def _with(board, cell, v):
    b = board.copy().reshape(-1)
    b[cell] = v
    return b.reshape(board.shape)


def test_self_play_light_weight(n):  # tests.rs:185-204 and async_agent/tests.rs:7-47
    res = o.self_play(n, num=4, threads=4, num_simulation=2, seed=3)
    assert len(res) == 4
    for r in res:
        assert r["err"] == o.OK and len(r["path"]) > 0
        if r["winner"] != Z:
            assert r["path"][-1][0] == r["winner"]
        # boards in the path replay exactly (agent_impl/mod.rs:139-145)
        g = o.new_game(n)
        for t, b, (row, col) in r["path"]:
            assert t == g.turn and (b == g.np_board()).all()
            assert o.game_set(g, row, col)[0] == o.OK


def test_self_play_is_reproducible_and_thread_independent():
    a = o.self_play(9, num=3, threads=1, num_simulation=8, seed=5, want_pi=True)
    b = o.self_play(9, num=3, threads=3, num_simulation=8, seed=5, want_pi=True)
    for x, y in zip(a, b):
        assert x["winner"] == y["winner"] and [p[2] for p in x["path"]] == [p[2] for p in y["path"]]
        assert (x["pi"] == y["pi"]).all()
        # the stone that fills the board has one child, a terminal leaf: expand returns before the visit increment
        # (mod.rs:338-345), so its pi is all zero; every other move has visited children
        assert (x["pi"].sum(axis=1)[:80] > 0).all()
    c = o.self_play(9, num=1, threads=1, num_simulation=8, seed=6)
    assert [p[2] for p in c[0]["path"]] != [p[2] for p in a[0]["path"]]


def test_transposition_merge_and_prune():
    """the tree is a table keyed by the board; the two same-colour stones of a turn
    transpose, the last expander's prior wins (mod.rs:366-370), and next() drops the root's level."""
    n = 9
    g = o.new_game(n)
    o.game_set(g, 4, 4)  # black; white now has two stones to place
    s = o.sim_from_game(g)
    az = o.AlphaZero(n, eval_kind=o.EVAL_RANDOM, seed=1, num_simulation=60)
    az.init(s)
    for _ in range(60):
        az.search(s)
    root = g.np_board()
    assert az.node_info(root).visit == 60
    # a white pair placed in either order is ONE node
    vis, q, p = az.children(root)
    expanded = [c for c in range(n * n) if vis[c] > 0]
    assert len(expanded) >= 2
    a, b = expanded[0], expanded[1]
    ba, bb = _with(root, a, W), _with(root, b, W)
    bab = _with(ba, b, W)
    info = az.node_info(bab)
    assert info.found and info.num_player == 3
    pa, pb = az.node_prob(ba), az.node_prob(bb)
    assert info.n_prob in (pa.reshape(-1)[b], pb.reshape(-1)[a])
    # map holds the root, its children and every child of every expanded node, merged
    assert az.map_len() < 1 + 60 * n * n
    az.prune(s)
    assert az.node_info(root).found           # root re-inserted (mod.rs:486)
    assert az.node_info(bab).found            # deeper levels persist = tree reuse
    other_level1 = _with(np.zeros((n, n), np.int8), 0, B)
    assert not az.node_info(other_level1).found


def test_one_level_backup():
    """the leaf value reaches q_sum of the immediate parent only (mod.rs:386-408)."""
    n = 9
    g = o.new_game(n)
    s = o.sim_from_game(g)
    az = o.AlphaZero(n, eval_kind=o.EVAL_RANDOM, seed=2, num_simulation=300)
    az.init(s)
    for _ in range(300):
        az.search(s)
    root = g.np_board()
    rinfo = az.node_info(root)
    vis, q, p = az.children(root)
    # root.q_sum == sum of value over expanded depth-1 children (each evaluated once as a leaf)
    tot = np.float32(0)
    depth1 = 0
    for c in range(n * n):
        if vis[c] > 0:
            depth1 += 1
    assert rinfo.visit == 300
    assert sum(max(v, 0) for v in vis) >= 299 - 0  # every search but the first passes through a child
    assert az.mean_depth() > 0.9
    assert abs(rinfo.q_value - rinfo.q_sum / rinfo.visit) < 1e-6
