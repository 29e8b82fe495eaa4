"""The reference's C ABI (cpp_self_play / cpp_play, N = 15) served by the GPU engine.
Reference tests mirrored: test_cppbind/main.cpp:46-57 (K-11) and the deterministic cpp_play vectors."""
import json
import os
import shutil
import subprocess
import numpy as np
import pytest

import c6oracle as o
from gpu_common import need_gpu, pkg

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = 15


def random_policy(turn, boards):  # Connect6/sample/python/random_policy.py:6-20 shape
    rng = np.random.default_rng(abs(hash(boards.tobytes())) % (2**32))
    n = boards.shape[0]
    return rng.uniform(-1, 1, n), rng.uniform(-1, 1, (n, N * N))


def check_result(res):
    for winner, path in res:
        assert len(path) > 0
        g = o.new_game(N)
        for turn, board, (r, c) in path:
            assert turn == g.turn and (board == g.np_board()).all()
            assert o.game_set(g, r, c)[0] == o.OK
        import ctypes as C
        assert o.lib().c6o_game_is_end(C.byref(g)) == winner
        if winner != 0:
            assert path[-1][0] == winner


def test_self_play_two_games():  # test_cppbind/main.cpp:46-52
    need_gpu()
    seen = []

    def policy(turn, boards):
        assert turn in (-1, 1) and boards.shape[0] % 8 == 0 and set(np.unique(boards)) <= {-1, 0, 1}
        seen.append(boards.shape[0])
        return random_policy(turn, boards)

    res = pkg.self_play_np(policy, {"num_simulation": 2, "num_game_thread": 2})
    assert len(res) == 2 and seen
    check_result(res)


def test_self_play_single_game_more_sims():
    need_gpu()
    res = pkg.self_play_np(random_policy, {"num_simulation": 12, "num_game_thread": 1})
    assert len(res) == 1
    check_result(res)


def test_play_main_policy():  # test_cppbind/main.cpp:31-44,54-57: last empty cell
    need_gpu()

    def main_policy(board):
        idx = np.flatnonzero(board.reshape(-1) == 0)
        return (int(idx[-1]) // N, int(idx[-1]) % N)

    res = pkg.play(main_policy, False, 2)
    assert len(res) == 2
    for winner, path in res:
        assert winner == -1 and len(path) == 81 and path[-1][2] == (9, 9)
    check_result(res)


def test_callback_evaluator_matches_oracle_with_same_callback():
    """same deterministic callback on both sides -> identical games (engine batches across games per side)"""
    need_gpu()
    n, G, S, seed = 9, 5, 14, 8

    def fn(player, boards):
        b = boards.astype(np.float32)
        k = np.arange(n * n, dtype=np.float32)
        pol = np.cos(b * 1.7 + k[None, :] * 0.37 + player) * 0.5
        val = np.tanh(b.sum(axis=1) * 0.1 + 0.05 * player)
        return val.astype(np.float32), pol.astype(np.float32)

    eng = pkg.Engine(board_size=n, num_games=G, num_simulation=S, seed=seed, evaluator=pkg.EVAL_CALLBACK)
    eng.set_callback(fn)
    eng.run()
    games = {g["uid"]: g for g in eng.fetch_games()}
    eng.close()

    def oracle_eval(ctx, turn, nn, ln, boards, values, policies):
        b = np.ctypeslib.as_array(boards, shape=(ln, nn * nn))
        v, p = fn(turn, b)
        np.ctypeslib.as_array(values, shape=(ln,))[:] = v
        np.ctypeslib.as_array(policies, shape=(ln, nn * nn))[:] = p
        return 0

    ref = o.self_play(n, num=G, threads=1, num_simulation=S, seed=seed, eval_kind=o.EVAL_CUSTOM, eval_fn=oracle_eval)
    for uid in range(G):
        mv = [r * n + c for _, _, (r, c) in ref[uid]["path"]]
        assert list(games[uid]["moves"]) == mv and games[uid]["winner"] == ref[uid]["winner"]


def test_apply_moves_engine_black_vs_outside_white():
    """c6_engine_apply_moves (MultiPolicy / cpp_play_with's human side): the engine searches and moves for Black, White's
    stones come from outside (first empty cell).  Bookkeeping (stone count, turn order 1-2-2, winner) must follow
    Game::set / is_game_end, and the first Black stone equals plain self-play's (same searches, same seed)."""
    need_gpu()
    import os
    n, S, seed = 9, 24, 11
    ref = pkg.Engine(board_size=n, num_games=1, num_simulation=S, seed=seed, evaluator=pkg.EVAL_RANDOM)
    ref.run()
    first = int(ref.fetch_games()[0]["moves"][0])
    ref.close()
    e = pkg.Engine(board_size=n, num_games=1, num_simulation=S, seed=seed, evaluator=pkg.EVAL_RANDOM)
    stones, mirror = 0, np.zeros(n * n, np.int8)
    turn, remain = -1, 1
    while True:
        r = e.root(0)
        assert r["turn"] == turn and r["remain"] == remain and r["stones"] == stones
        if turn == -1:
            e.step(S)                                   # Black: S searches + the engine's own move
        else:
            e.apply_moves([int(np.flatnonzero(mirror == 0)[0])])
        r = e.root(0)
        board = r["board"].reshape(-1)
        new = np.flatnonzero(board != mirror)
        if r["status"] == 0 or r["winner"] != 0:
            assert new.size == 1 and board[new[0]] == turn
            if stones == 0:
                assert new[0] == first
            mirror = board.copy()
            stones += 1
        remain -= 1
        if remain <= 0:
            remain, turn = 2, -turn
        if r["status"] != 0:
            break
    assert r["status"] == 1 and r["winner"] == o.search(mirror.reshape(n, n))
    assert e.stats()["overflow_games"] == 0
    e.close()


def test_cpp_play_with_scripted_human(tmp_path):
    """cpp_play_with (lib.rs:140-169): AlphaZero(Black, host callback) against scripted stdin moves; run in a subprocess
    because the human side reads the process's stdin.  White walks along a row; a collision with a Black stone makes the
    library abort like the reference's unwrap(), so a few rows are tried."""
    need_gpu()
    import subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prog = textwrap.dedent(f"""
        import sys, numpy as np
        sys.path.insert(0, {root!r})
        from c6amd_loader import load_package
        pkg = load_package()
        def policy(turn, boards):
            k = len(boards)
            assert isinstance(boards, list) and len(boards[0]) == 225
            return np.zeros(k, np.float32), np.full((k, 225), 1 / 225, np.float32)
        param = pkg.default_param()
        param["num_simulation"] = 8
        winner, path = pkg.play_with(policy, param)
        assert isinstance(path[0][1], list) and len(path[0][1]) == 225
        print("RESULT", winner, len(path), path[-1][0])
    """)
    ok = False
    for row in "onmlk":
        script = "xx\n" + "".join(f"{row}{c}\n" for c in "ABCDEFGH")       # first line: invalid input, retried
        r = subprocess.run([sys.executable, "-c", prog], input=script, capture_output=True, text=True, timeout=300, env=dict(os.environ, C6_SEED="7"))
        if r.returncode != 0:
            assert "already set position" in r.stderr, r.stderr[-500:]
            continue
        out = r.stdout
        assert "invalid input, retry" in out and "0 A B C D E F G H I J K L M N O" in out and "White (" in out and "Black (" in out
        line = [l for l in out.splitlines() if l.startswith("RESULT")][0].split()
        assert int(line[1]) == 1 and int(line[3]) == 1 and int(line[2]) >= 11          # White wins with its sixth stone in the row
        ok = True
        break
    assert ok


def test_net_vs_net_two_engines_stay_in_step():
    """MultiPolicy with two resident nets (tools/net_vs_net.py): both engines must record the same games -- every stone
    searched by one side and applied from outside on the other -- and every game must end like the oracle's
    search says."""
    need_gpu()
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import net_vs_net
    import __graft_entry__ as ge
    n, C, nb, G, S = 9, 64, 2, 24, 6
    ba = ge.make_random_blob(n, C, nb, np.random.default_rng(1))
    bb = ge.make_random_blob(n, C, nb, np.random.default_rng(2))
    ga, gb = net_vs_net.play_match(pkg, n, G, S, ba, bb, C, nb, seed=5)
    assert len(ga) == len(gb) == G
    key = lambda g: g["uid"]
    for a, b in zip(sorted(ga, key=key), sorted(gb, key=key)):
        assert list(a["moves"]) == list(b["moves"]) and a["winner"] == b["winner"]
        board = np.zeros(n * n, np.int8)
        for turn, _, cell in pkg.replay_moves(n, a["moves"]):
            board[cell] = turn
        assert o.search(board.reshape(n, n)) == a["winner"]


def test_cpp_play_with_random_policy():
    """RandomPolicy (policy/random_policy/mod.rs:39-47) behind cpp_play: complete legal games, winner = the oracle's
    search on the final position, last mover = winner (agent_impl/tests.rs invariant)"""
    need_gpu()
    res = pkg.play(pkg.RandomPolicy(seed=3), num_game_thread=16)
    assert len(res) == 16
    for winner, path in res:
        board = path[-1][1].copy()
        turn, _, (r, c) = path[-1]
        assert board[r, c] == 0
        board[r, c] = turn
        assert o.search(board) == winner
        if winner != 0:
            assert turn == winner
        turns = [t for t, _, _ in path]
        assert turns[0] == -1 and turns[1:3] == [1, 1]          # 1 stone, then 2 per turn


# ---- the Python face with the reference's own objects ------------------------------

def _check_ref_result(winner, path):
    """(winner: int, [(turn: int, board: list[225] of int, (row, col))]) -- pybind_impl/mod.rs:66-105 -- replayed on the oracle"""
    assert isinstance(winner, int) and isinstance(path, list) and len(path) > 0
    g = o.new_game(N)
    for turn, board, pos in path:
        assert isinstance(turn, int) and isinstance(board, list) and len(board) == N * N and all(isinstance(x, int) for x in board)
        assert isinstance(pos, tuple) and len(pos) == 2
        assert turn == g.turn and board == g.np_board().reshape(-1).tolist()
        assert o.game_set(g, pos[0], pos[1])[0] == o.OK
    import ctypes as C
    assert o.lib().c6o_game_is_end(C.byref(g)) == winner
    if winner != 0:
        assert path[-1][0] == winner


@pytest.mark.parametrize("threads", [1, 2])
def test_python_face_runs_the_reference_sample(threads, capfd):
    """The body of Connect6/sample/python/random_policy.py:6-37 with `pyconnect6` = this package: class RandomPolicy built on
    `board_size()`, called with `(turn, board)` where `board` is a list of `len` lists of N*N ints, returning numpy arrays;
    `winner, path = self_play(policy, param)` with num_game_thread = 1 (lib.rs:119-128), a tuple of (winner, path) with 2
    (lib.rs:129-152); debug = True prints the boards like the reference's sample run."""
    need_gpu()
    pyconnect6 = pkg
    calls = []

    class RandomPolicy:
        def __init__(self):
            self.board_size = pyconnect6.board_size()

        def __call__(self, turn, board):
            assert isinstance(turn, int) and turn in (-1, 1)
            assert isinstance(board, list) and isinstance(board[0], list) and isinstance(board[0][0], int)
            assert len(board[0]) == self.board_size ** 2 and set(board[0]) <= {-1, 0, 1}
            calls.append(len(board))
            size = len(board)
            value = np.random.rand(size)
            rand_policy = np.random.rand(size, self.board_size * self.board_size)
            return value, rand_policy

    policy = RandomPolicy()
    param = pyconnect6.default_param()
    param['num_simulation'] = 2
    param['num_game_thread'] = threads
    param['debug'] = True
    if threads == 1:
        winner, path = pyconnect6.self_play(policy, param)
        print('winner {}, len {}'.format(winner, len(path)))
        _check_ref_result(winner, path)
        out = capfd.readouterr().out
        assert "0 A B C D E F G H I J K L M N O" in out and "Black (" in out and f"winner {winner}, len {len(path)}" in out
    else:
        res = pyconnect6.self_play(policy, param)
        assert isinstance(res, tuple) and len(res) == threads
        for winner, path in res:
            _check_ref_result(winner, path)
    assert calls and all(k % 8 == 0 for k in calls)          # always the 8 dihedral images (x games that share the side to move)


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_cpp_caller_self_play_with_a_host_callback(tmp_path):
    """The boundary from a compiled C++ caller (tools/cpp/dropin_rate.cpp: include/connect6_amd.h, `cpp_self_play` with an evaluator callback
    of the reference's shape, connect6.hpp:14 -- the reference's own sample is compiled in tests/test_abi_cpu.py, here one RUNS on the GPU):
    every game comes back, every simulation went through the callback (8 boards per game and round)."""
    need_gpu()
    exe = tmp_path / "dropin_rate"
    libdir = os.path.join(ROOT, "alphazero-connect6_amd")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "cpp", "dropin_rate.cpp"),
                           "-L" + libdir, "-lconnect6_amd", "-Wl,-rpath," + libdir, "-o", str(exe)])
    r = json.loads(subprocess.check_output([str(exe), "5", "6"], text=True, timeout=300).strip().splitlines()[-1])
    assert r["games"] == 5 and r["sims_per_move"] == 6 and r["stones"] >= 5 * 11 and r["callback_calls"] > 0
    assert r["boards_per_s"] > 0 and r["sims_per_s"] > 0
