"""ctypes binding of the CPU oracle (oracle/c6_oracle.h).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product never imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")
SO = os.path.join(ODIR, "_build", "libc6oracle.so")

MAXCELLS = 361
BLACK, NONE, WHITE = -1, 0, 1
OK, ERR_INVALID, ERR_ALREADY = 0, 1, 2
EVAL_CUSTOM, EVAL_RANDOM, EVAL_UNIFORM = 0, 1, 2


def build(force=False):
    # C6_ORACLE_SO: another build of the same sources (the AddressSanitizer / UBSan one of tools/oracle_sanitize.sh)
    if os.environ.get("C6_ORACLE_SO"):
        return os.environ["C6_ORACLE_SO"]
    srcs = [os.path.join(ODIR, f) for f in ("c6_oracle.c", "c6_oracle_net.c", "c6_default.c", "c6_oracle.h", "Makefile")]
    if force or not os.path.exists(SO) or any(os.path.getmtime(s) > os.path.getmtime(SO) for s in srcs):
        subprocess.check_call(["make", "-C", ODIR], stdout=subprocess.DEVNULL)
    return SO


class Game(C.Structure):
    _fields_ = [("n", C.c_int), ("turn", C.c_int), ("num_remain", C.c_int), ("board", C.c_int8 * MAXCELLS)]

    def np_board(self):
        return np.ctypeslib.as_array(self.board)[: self.n * self.n].copy().reshape(self.n, self.n)


class SetResult(C.Structure):
    _fields_ = [("player", C.c_int), ("num_remain", C.c_int), ("row", C.c_int), ("col", C.c_int)]


class Sim(C.Structure):
    _fields_ = [("n", C.c_int), ("turn", C.c_int), ("num_remain", C.c_int), ("npossible", C.c_int),
                ("possible", C.c_int16 * MAXCELLS), ("board", C.c_int8 * MAXCELLS)]

    def np_board(self):
        return np.ctypeslib.as_array(self.board)[: self.n * self.n].copy().reshape(self.n, self.n)

    def possible_list(self):
        return [int(self.possible[i]) for i in range(self.npossible)]


class Hyper(C.Structure):
    _fields_ = [("num_simulation", C.c_int), ("epsilon", C.c_float), ("dirichlet_alpha", C.c_double),
                ("c_puct", C.c_float)]


class NodeInfo(C.Structure):
    _fields_ = [("found", C.c_int), ("visit", C.c_int), ("value", C.c_float), ("q_sum", C.c_float),
                ("q_value", C.c_float), ("n_prob", C.c_float), ("num_player", C.c_int), ("n_next", C.c_int),
                ("has_prob", C.c_int)]


class Path(C.Structure):
    _fields_ = [("turn", C.c_int), ("row", C.c_int), ("col", C.c_int), ("board", C.c_int8 * MAXCELLS)]


class PlayResult(C.Structure):
    _fields_ = [("winner", C.c_int), ("len", C.c_int), ("err", C.c_int), ("path", C.POINTER(Path)),
                ("pi", C.POINTER(C.c_int32))]


EVAL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int8), C.POINTER(C.c_float),
                      C.POINTER(C.c_float))
POLICY_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(Game), C.POINTER(C.c_int), C.POINTER(C.c_int))

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build())
    i8p, f32p, f64p, i32p = (C.POINTER(C.c_int8), C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int))
    L.c6o_game_new.argtypes = [C.POINTER(Game), C.c_int]
    L.c6o_game_set.argtypes = [C.POINTER(Game), C.c_int, C.c_int, C.POINTER(SetResult)]
    L.c6o_game_set.restype = C.c_int
    L.c6o_game_is_end.argtypes = [C.POINTER(Game)]
    L.c6o_strerror.restype = C.c_char_p
    L.c6o_strerror.argtypes = [C.c_int]
    L.c6o_search.argtypes = [i8p, C.c_int]
    L.c6o_sim_from_game.argtypes = [C.POINTER(Sim), C.POINTER(Game)]
    L.c6o_sim_validate.argtypes = [C.POINTER(Sim), C.c_int, C.c_int]
    L.c6o_sim_next_turn.argtypes = [C.POINTER(Sim)]
    L.c6o_sim_simulate_in.argtypes = [C.POINTER(Sim), C.c_int, C.c_int]
    L.c6o_sim_rollback_in.argtypes = [C.POINTER(Sim), C.c_int, C.c_int]
    L.c6o_diff_board.argtypes = [i8p, i8p, C.c_int]
    for nm in ("rotate_left", "rotate_right", "flip_vertical", "flip_horizontal"):
        getattr(L, f"c6o_{nm}_i8").argtypes = [i8p, C.c_int]
    L.c6o_augment_way8.argtypes = [i8p, C.c_int, i8p]
    L.c6o_recover_way8.argtypes = [f32p, C.c_int, f32p]
    L.c6o_philox4x32_10.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.c6o_log4.restype = C.c_float
    L.c6o_log4.argtypes = [C.c_float]
    L.c6o_exp4.restype = C.c_float
    L.c6o_exp4.argtypes = [C.c_float, C.POINTER(C.c_int)]
    L.c6o_gamma_draw4.argtypes = [C.c_double, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_int)]
    L.c6o_gamma_draw.restype = C.c_double
    L.c6o_gamma_draw.argtypes = [C.c_double, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    L.c6o_butterfly_sum64f.restype = C.c_float
    L.c6o_butterfly_sum64f.argtypes = [C.POINTER(C.c_float)]
    L.c6o_dirichlet_noise.argtypes = [C.c_double, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, i8p, C.c_int, f32p]
    L.c6o_az_new.restype = C.c_void_p
    L.c6o_az_new.argtypes = [C.c_int, C.POINTER(Hyper), C.c_int, EVAL_FN, C.c_void_p, C.c_uint64, C.c_uint32]
    L.c6o_az_free.argtypes = [C.c_void_p]
    L.c6o_az_next.argtypes = [C.c_void_p, C.POINTER(Game)]
    for nm in ("init", "select", "expand", "search", "policy", "prune"):
        getattr(L, f"c6o_az_{nm}").argtypes = [C.c_void_p, C.POINTER(Sim)]
    L.c6o_az_update.argtypes = [C.c_void_p, C.POINTER(Sim), C.POINTER(C.c_int16), C.c_int]
    L.c6o_az_map_len.restype = C.c_size_t
    L.c6o_az_map_len.argtypes = [C.c_void_p]
    L.c6o_az_search_count.restype = C.c_uint32
    L.c6o_az_search_count.argtypes = [C.c_void_p]
    L.c6o_az_mean_depth.restype = C.c_double
    L.c6o_az_mean_depth.argtypes = [C.c_void_p]
    L.c6o_az_node_info.argtypes = [C.c_void_p, i8p, C.POINTER(NodeInfo)]
    L.c6o_az_node_prob.argtypes = [C.c_void_p, i8p, f32p]
    L.c6o_az_children.argtypes = [C.c_void_p, i8p, i32p, f32p, f32p]
    L.c6o_agent_play.argtypes = [C.c_int, POLICY_FN, C.c_void_p, C.POINTER(PlayResult)]
    L.c6o_play_result_free.argtypes = [C.POINTER(PlayResult)]
    L.c6o_self_play.argtypes = [C.c_int, C.POINTER(Hyper), C.c_int, EVAL_FN, C.c_void_p, C.c_uint64, C.c_uint32,
                                C.c_int, C.c_int, C.c_int, C.POINTER(PlayResult)]
    L.c6o_replay_new.restype = C.c_void_p
    L.c6o_replay_new.argtypes = [C.c_int, C.c_size_t]
    L.c6o_replay_free.argtypes = [C.c_void_p]
    L.c6o_replay_push_game.argtypes = [C.c_void_p, C.c_int, C.POINTER(Path), C.c_int]
    L.c6o_replay_clear.argtypes = [C.c_void_p]
    L.c6o_replay_size.restype = C.c_size_t
    L.c6o_replay_size.argtypes = [C.c_void_p]
    L.c6o_replay_tail.restype = C.c_size_t
    L.c6o_replay_tail.argtypes = [C.c_void_p]
    L.c6o_replay_sample.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_uint32, f32p, f32p, f32p, f32p]
    L.c6o_net_blob_floats.restype = C.c_size_t
    L.c6o_net_blob_floats.argtypes = [C.c_int, C.c_int, C.c_int]
    L.c6o_net_new.restype = C.c_void_p
    L.c6o_net_new.argtypes = [C.c_int, C.c_int, C.c_int, f32p]
    L.c6o_net_free.argtypes = [C.c_void_p]
    L.c6o_net_forward.argtypes = [C.c_void_p, C.c_float, f32p, C.c_int, f32p, f32p]
    L.c6o_net_eval.restype = C.c_int
    L.c6o_net_eval.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, i8p, f32p, f32p]
    L.c6o_default_new.restype = C.c_void_p
    L.c6o_default_new.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_uint32]
    L.c6o_default_free.argtypes = [C.c_void_p]
    for nm in ("init", "select", "expand", "policy"):
        getattr(L, f"c6o_default_{nm}").argtypes = [C.c_void_p, C.POINTER(Sim)]
    L.c6o_default_update.argtypes = [C.c_void_p, C.POINTER(Sim), C.POINTER(C.c_int16), C.c_int]
    L.c6o_default_search.argtypes = [C.c_void_p, C.POINTER(Game)]
    L.c6o_default_next.argtypes = [C.c_void_p, C.POINTER(Game)]
    L.c6o_default_map_len.restype = C.c_size_t
    L.c6o_default_map_len.argtypes = [C.c_void_p]
    L.c6o_default_search_count.restype = C.c_uint32
    L.c6o_default_search_count.argtypes = [C.c_void_p]
    L.c6o_default_node.argtypes = [C.c_void_p, i8p, i32p, i32p, i32p, C.c_int, i32p, i32p, i32p]
    _lib = L
    return L


def i8(a):
    return a.ctypes.data_as(C.POINTER(C.c_int8))


def f32(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def search(board):
    b = np.ascontiguousarray(board, dtype=np.int8)
    return lib().c6o_search(i8(b), b.shape[0])


def new_game(n):
    g = Game()
    lib().c6o_game_new(C.byref(g), n)
    return g


def game_set(g, row, col):
    r = SetResult()
    err = lib().c6o_game_set(C.byref(g), row, col, C.byref(r))
    return err, r


def sim_from_game(g):
    s = Sim()
    lib().c6o_sim_from_game(C.byref(s), C.byref(g))
    return s


def augment_way8(board):
    b = np.ascontiguousarray(board, dtype=np.int8)
    n = b.shape[0]
    out = np.zeros((8, n, n), np.int8)
    lib().c6o_augment_way8(i8(b), n, i8(out))
    return out


def recover_way8(probs8):
    p = np.array(probs8, dtype=np.float32, order="C")
    n = p.shape[1]
    out = np.zeros((n, n), np.float32)
    lib().c6o_recover_way8(f32(p), n, f32(out))
    return out


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().c6o_philox4x32_10(c, k, o)
    return [int(x) for x in o]


def dirichlet_noise(alpha, k0, k1, search_idx, depth, board):
    b = np.ascontiguousarray(board, dtype=np.int8)
    n = b.shape[0]
    out = np.zeros((n, n), np.float32)
    lib().c6o_dirichlet_noise(alpha, k0, k1, search_idx, depth, i8(b), n, f32(out))
    return out


NULL_EVAL = C.cast(None, EVAL_FN)


class AlphaZero:
    """Handle on c6o_az (policy/alphazero_policy/mod.rs AlphaZero)."""

    def __init__(self, n, num_simulation=800, epsilon=0.25, alpha=0.03, c_puct=1.0, eval_kind=EVAL_RANDOM,
                 eval_fn=None, seed=0, game_uid=0, ctx=None):
        self.n = n
        self.hp = Hyper(num_simulation, epsilon, alpha, c_puct)
        self._fn = EVAL_FN(eval_fn) if (eval_fn is not None and not isinstance(eval_fn, EVAL_FN)) else eval_fn
        fn = self._fn if self._fn is not None else NULL_EVAL
        self.h = lib().c6o_az_new(n, C.byref(self.hp), eval_kind, fn, ctx, seed, game_uid)

    def close(self):
        if self.h:
            lib().c6o_az_free(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def next(self, g):
        return lib().c6o_az_next(self.h, C.byref(g))

    def init(self, s):
        lib().c6o_az_init(self.h, C.byref(s))

    def select(self, s):
        return lib().c6o_az_select(self.h, C.byref(s))

    def expand(self, s):
        lib().c6o_az_expand(self.h, C.byref(s))

    def update(self, s, path):
        arr = (C.c_int16 * max(1, len(path)))(*path)
        lib().c6o_az_update(self.h, C.byref(s), arr, len(path))

    def search(self, s):
        lib().c6o_az_search(self.h, C.byref(s))

    def policy(self, s):
        return lib().c6o_az_policy(self.h, C.byref(s))

    def prune(self, s):
        lib().c6o_az_prune(self.h, C.byref(s))

    def map_len(self):
        return lib().c6o_az_map_len(self.h)

    def mean_depth(self):
        return lib().c6o_az_mean_depth(self.h)

    def node_info(self, board):
        b = np.ascontiguousarray(board, dtype=np.int8)
        info = NodeInfo()
        lib().c6o_az_node_info(self.h, i8(b), C.byref(info))
        return info

    def node_prob(self, board):
        b = np.ascontiguousarray(board, dtype=np.int8)
        out = np.zeros(self.n * self.n, np.float32)
        ok = lib().c6o_az_node_prob(self.h, i8(b), f32(out))
        return out.reshape(self.n, self.n) if ok else None

    def children(self, board):
        b = np.ascontiguousarray(board, dtype=np.int8)
        cells = self.n * self.n
        vis = np.zeros(cells, np.int32)
        q = np.zeros(cells, np.float32)
        p = np.zeros(cells, np.float32)
        lib().c6o_az_children(self.h, i8(b), vis.ctypes.data_as(C.POINTER(C.c_int)), f32(q), f32(p))
        return vis, q, p


class DefaultPolicy:
    """policy/default_policy/mod.rs:84-275 (pure MCTS with random playouts) on the C6RNG stream."""

    def __init__(self, n, num_iter=50, seed=0, uid=0):
        self.n, self.L = n, lib()
        self.h = self.L.c6o_default_new(n, num_iter, seed, uid)

    def close(self):
        if self.h:
            self.L.c6o_default_free(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def next(self, g):
        """cell (row * n + col) or -1"""
        return self.L.c6o_default_next(self.h, C.byref(g))

    def init(self, s):
        self.L.c6o_default_init(self.h, C.byref(s))

    def select(self, s):
        return self.L.c6o_default_select(self.h, C.byref(s))

    def expand(self, s):
        return self.L.c6o_default_expand(self.h, C.byref(s))

    def update(self, s, path):
        arr = (C.c_int16 * max(len(path), 1))(*path)
        self.L.c6o_default_update(self.h, C.byref(s), arr, len(path))

    def search(self, g):
        self.L.c6o_default_search(self.h, C.byref(g))

    def policy(self, s):
        return self.L.c6o_default_policy(self.h, C.byref(s))

    def map_len(self):
        return self.L.c6o_default_map_len(self.h)

    def search_count(self):
        return self.L.c6o_default_search_count(self.h)

    def node(self, board, cap=1024):
        """None, or (visit, black_win, [(cell, visit, black_win) of the children in push order])"""
        b = np.ascontiguousarray(board, dtype=np.int8).reshape(-1)
        v, w, k = C.c_int(), C.c_int(), C.c_int()
        cc, cv, cw = (np.zeros(cap, np.int32) for _ in range(3))
        i32p = C.POINTER(C.c_int)
        if not self.L.c6o_default_node(self.h, i8(b), C.byref(v), C.byref(w), C.byref(k), cap, cc.ctypes.data_as(i32p),
                                       cv.ctypes.data_as(i32p), cw.ctypes.data_as(i32p)):
            return None
        m = min(k.value, cap)
        return v.value, w.value, [(int(cc[i]), int(cv[i]), int(cw[i])) for i in range(m)]


def _result_to_py(r, n, want_pi=False):
    cells = n * n
    paths = []
    for i in range(r.len):
        p = r.path[i]
        board = np.ctypeslib.as_array(p.board)[:cells].copy().reshape(n, n)
        paths.append((p.turn, board, (p.row, p.col)))
    pi = None
    if want_pi and r.len and bool(r.pi):
        pi = np.ctypeslib.as_array(r.pi, shape=(r.len, cells)).copy()
    out = dict(winner=r.winner, err=r.err, path=paths, pi=pi)
    lib().c6o_play_result_free(C.byref(r))
    return out


def agent_play(n, policy):
    """policy(game) -> (row, col) or None.  agent/agent_impl/mod.rs:121-178"""

    def tramp(ctx, gptr, rowp, colp):
        pos = policy(gptr.contents)
        if pos is None:
            return 0
        rowp[0], colp[0] = int(pos[0]), int(pos[1])
        return 1

    cb = POLICY_FN(tramp)
    r = PlayResult()
    lib().c6o_agent_play(n, cb, None, C.byref(r))
    return _result_to_py(r, n)


def self_play(n, num=1, threads=1, num_simulation=2, epsilon=0.25, alpha=0.03, c_puct=1.0, eval_kind=EVAL_RANDOM,
              eval_fn=None, ctx=None, seed=0, uid_base=0, want_pi=False):
    hp = Hyper(num_simulation, epsilon, alpha, c_puct)
    fn = eval_fn if eval_fn is not None else NULL_EVAL
    if eval_fn is not None and not isinstance(eval_fn, EVAL_FN):
        fn = EVAL_FN(eval_fn)
    res = (PlayResult * num)()
    lib().c6o_self_play(n, C.byref(hp), eval_kind, fn, ctx, seed, uid_base, num, threads, int(want_pi), res)
    return [_result_to_py(res[i], n, want_pi) for i in range(num)]


class Net:
    """oracle FP32 net (AlphaZero/cpp_weighted/model.hpp)."""

    def __init__(self, n, channels, num_block, blob):
        blob = np.ascontiguousarray(blob, dtype=np.float32)
        assert blob.size == lib().c6o_net_blob_floats(n, channels, num_block), (blob.size, lib().c6o_net_blob_floats(n, channels, num_block))
        self.n, self.h = n, lib().c6o_net_new(n, channels, num_block, f32(blob))

    def forward(self, player, boards):
        b = np.ascontiguousarray(boards, dtype=np.float32).reshape(-1, self.n * self.n)
        pol = np.zeros_like(b)
        val = np.zeros(b.shape[0], np.float32)
        lib().c6o_net_forward(self.h, float(player), f32(b), b.shape[0], f32(pol), f32(val))
        return pol, val

    def eval_fn(self):
        """(EVAL_FN pointer to c6o_net_eval, ctx) usable as a custom evaluator."""
        fn = C.cast(lib().c6o_net_eval, EVAL_FN)
        return fn, self.h

    def close(self):
        if self.h:
            lib().c6o_net_free(self.h)
            self.h = None


class Replay:
    """oracle replay buffer (AlphaZero/cpp_weighted/ReplayBuffer.hpp)"""

    def __init__(self, n, max_size):
        self.n, self.h = n, lib().c6o_replay_new(n, max_size)

    def push_game(self, winner, path):
        """path: [(turn, board[n,n], (row, col)), ...]"""
        arr = (Path * max(1, len(path)))()
        for i, (turn, board, (row, col)) in enumerate(path):
            arr[i].turn, arr[i].row, arr[i].col = int(turn), int(row), int(col)
            flat = np.ascontiguousarray(board, dtype=np.int8).reshape(-1)
            C.memmove(arr[i].board, flat.ctypes.data, flat.size)
        lib().c6o_replay_push_game(self.h, int(winner), arr, len(path))

    def clear(self):
        lib().c6o_replay_clear(self.h)

    def size(self):
        return lib().c6o_replay_size(self.h), lib().c6o_replay_tail(self.h)

    def sample(self, num_sample, seed, draw_id=0):
        cells = self.n * self.n
        w = np.zeros(num_sample, np.float32); p = np.zeros(num_sample, np.float32)
        b = np.zeros((num_sample, cells), np.float32); q = np.zeros(num_sample, np.float32)
        lib().c6o_replay_sample(self.h, num_sample, seed, draw_id, f32(w), f32(p), f32(b), f32(q))
        return w, p, b, q

    def close(self):
        if self.h:
            lib().c6o_replay_free(self.h)
            self.h = None
