"""CPU-side checks of the drop-in boundary (no compute, no GPU): the library loads, exports every symbol that
include/connect6_amd.h declares, has the reference's struct layouts, and the host-only FFI echo hooks reproduce
the reference's cross-language tests K-9/K-10 (cppbind/test_cppbind/rawobj.cpp, cppeval.cpp, cpp_policy.cpp;
Rust twins cppbind/.../rawobj/tests.rs, cppeval/tests.rs, cpp_policy/tests.rs)."""
import ctypes as C
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

from c6amd_loader import load_package

pkg = load_package()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = 15


def test_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "connect6_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b((?:c6_|cpp_|test_)[a-z0-9_]+)\s*\(", hdr))
    names -= {"c6_callback", "c6_policy_callback"}
    assert {"cpp_self_play", "cpp_play", "cpp_play_with", "c6_engine_create", "c6_board_step", "test_cpp_policy"} <= names
    L = pkg.lib()
    missing = [n for n in sorted(names) if not hasattr(L, n)]
    assert not missing, missing
    assert set(pkg.DROPIN_SYMBOLS + pkg.ENGINE_SYMBOLS) <= names


def test_struct_layouts_match_reference():  # 912 / 24 / 16 bytes, board @4, row @904, col @908
    assert C.sizeof(pkg.Path) == 912 and pkg.Path.board.offset == 4 and pkg.Path.row.offset == 904 and pkg.Path.col.offset == 908
    assert C.sizeof(pkg.PlayResult) == 24 and pkg.PlayResult.paths.offset == 8 and pkg.PlayResult.len.offset == 16
    assert C.sizeof(pkg.Vec) == 16


def test_no_gpu_fails_loudly():
    if pkg.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.C6Error, match="no HIP device"):
        pkg.Engine(board_size=9)
    with pytest.raises(pkg.C6Error):
        pkg.search_winner(9, np.zeros((81, 4), np.int8))
    with pytest.raises(pkg.C6Error, match="no HIP device"):       # the trainer has no CPU fallback either
        pkg.Trainer(9, 64, 2, 8)
    with pytest.raises(pkg.C6Error, match="no gfx950 device"):    # nor does the multi-device group behind the C ABI (csrc/group.cpp)
        pkg.Group(devices=[0, 0], board_size=9, num_games=4)


def test_group_shard_plan_matches_the_python_hosts():
    """c6_group_create's shard plan (csrc/group.cpp) restated here against alphazero-connect6_amd/shard.py, the plan bench.py's ranks use: member r of n gets
    shard_games(total, n, r) slots starting at uid `first`, and a recycled slot advances by the total -- for equal shards exactly uid_plan's (base, stride)."""
    from alphazero_connect6_amd import shard
    for total, n in ((12, 2), (7, 3), (65536, 8), (4096, 1), (100, 6)):
        per, rem = total // n, total % n
        first = 0
        for r in range(n):
            mine = per + (1 if r < rem else 0)                  # group.cpp: mb.games
            f = r * per + min(r, rem)                           # group.cpp: mb.first_uid
            assert (mine, f) == shard.shard_games(total, n, r) and f == first
            if rem == 0:
                assert shard.uid_plan(per, n, r) == (f, total)  # (base, stride): c.game_uid_base, c.game_uid_stride = cfg->num_games
            first += mine
        assert first == total


def test_raw_path_hooks():  # rawobj.cpp:5-65
    L = pkg.lib()
    p = L.test_new_raw_path()
    assert (p.turn, p.row, p.col) == (0, 0, 0) and not np.ctypeslib.as_array(p.board).any()
    p = L.test_with_raw_path()
    assert (p.turn, p.row, p.col) == (1, 0, N % 5 + 1)
    want = (np.arange(N * N) % 3 - 1).reshape(N, N)
    assert (np.ctypeslib.as_array(p.board) == want).all()
    rng = np.random.default_rng(0)
    board = rng.integers(-1, 2, size=(N, N)).astype(np.int32)
    p = L.test_echo_raw_path(-1, board.ctypes.data_as(C.POINTER(C.c_int)), 3, 7)
    assert (p.turn, p.row, p.col) == (-1, 3, 7) and (np.ctypeslib.as_array(p.board) == board).all()


def test_raw_play_result_hooks():  # rawobj.cpp:67-140, ffi_test/mod.rs:60-121
    L = pkg.lib()
    arena = pkg._Arena()
    r = L.test_with_raw_play_result(arena.alloc_path)
    assert r.winner == -1 and r.len == 10
    player = -1
    for i in range(10):
        p = r.paths[i]
        b = np.ctypeslib.as_array(p.board)
        want = np.zeros((N, N), np.int32)
        for j in range(i + 1):
            want[j, j] = (i + j) % 3 - 1
        assert (b == want).all() and (p.turn, p.row, p.col) == (player, i, i + 1)
        player = -player
    e = L.test_echo_raw_play_result(1, r.paths, r.len, arena.alloc_path)
    assert e.winner == 1 and e.len == 10
    for i in range(10):
        assert (np.ctypeslib.as_array(e.paths[i].board) == np.ctypeslib.as_array(r.paths[i].board)).all()
        assert (e.paths[i].turn, e.paths[i].row, e.paths[i].col) == (r.paths[i].turn, r.paths[i].row, r.paths[i].col)


def test_raw_vec_hooks():  # rawobj.cpp:142-171
    L = pkg.lib()
    arena = pkg._Arena()
    v = L.test_with_raw_vec(arena.alloc_int)
    assert [v.vec[i] for i in range(v.len)] == [0, 1, 2, 3, 4, 5]
    src = (C.c_int * 7)(5, 4, 3, 2, 1, 0, -9)
    v = L.test_echo_raw_vec(src, 7, arena.alloc_int)
    assert [v.vec[i] for i in range(v.len)] == [5, 4, 3, 2, 1, 0, -9]


def test_echo_cppeval():  # cppeval.cpp:5-61 and cppeval/tests.rs:6-92: values[i] = i + player, policies *= 2
    L = pkg.lib()
    arena = pkg._Arena()

    def cb(player, values, policies, n):
        for i in range(n):
            values[i] = i + player
        pol = np.ctypeslib.as_array(policies, shape=(n * N * N,))
        assert set(np.unique(pol)) <= {-1.0, 0.0, 1.0}      # board_to_float, cpp_support.rs:10-18
        pol *= 2

    rng = np.random.default_rng(1)
    ln, turn = 37, -1
    boards = rng.integers(-1, 2, size=ln * N * N).astype(np.int32)
    res = L.test_echo_cppeval(turn, boards.ctypes.data_as(C.POINTER(C.c_int)), ln, pkg.CALLBACK(cb), arena.alloc_float)
    assert res.len == ln + ln * N * N
    out = np.ctypeslib.as_array(res.vec, shape=(res.len,))
    assert (out[:ln] == np.arange(ln) + turn).all()
    assert (out[ln:] == boards * 2).all()


def test_cpp_policy_hook():  # cpp_policy.cpp:8-46
    L = pkg.lib()
    arena = pkg._Arena()
    rng = np.random.default_rng(2)
    board = rng.integers(-1, 2, size=N * N).astype(np.float32)
    seen = {}

    def cb(b, position):
        seen["board"] = np.ctypeslib.as_array(b, shape=(N * N,)).copy()
        position[0], position[1] = 1000, 1023

    v = L.test_cpp_policy(board.ctypes.data_as(C.POINTER(C.c_float)), pkg.POLICY_CALLBACK(cb), arena.alloc_int)
    assert v.len == 2 and (v.vec[0], v.vec[1]) == (1000, 1023) and (seen["board"] == board).all()
    v = L.test_cpp_policy(board.ctypes.data_as(C.POINTER(C.c_float)), pkg.POLICY_CALLBACK(lambda b, p: None), arena.alloc_int)
    assert v.len == 0                                        # position stays (-1, -1) -> "no move"


def test_tuple_stride_and_decode():
    L = pkg.lib()
    assert L.c6_tuple_stride(19) == 1088 and L.c6_tuple_stride(15) == 680 and L.c6_tuple_stride(9) == 248
    n, cells = 9, 81
    rec = np.zeros((2, 248), np.uint8)
    rec[0, :cells] = np.int8(-1).view(np.uint8)
    rec[0, 82:82 + 162] = np.arange(cells, dtype=np.uint16).view(np.uint8)
    rec[0, -2], rec[0, -1] = np.int8(1).view(np.uint8), np.int8(-1).view(np.uint8)
    d = pkg.decode_tuples(rec, n)
    assert (d["state"][0] == -1).all() and (d["pi"][0] == np.arange(cells)).all() and d["turn"][0] == 1 and d["z"][0] == -1


REF_SAMPLE = "/root/reference/Connect6/sample/cpp/main.cpp"


@pytest.mark.skipif(not os.path.exists(REF_SAMPLE) or shutil.which("g++") is None,
                    reason="the reference checkout exists only in the build container")
def test_reference_sample_links(tmp_path):
    """ABI conformance: the reference's own C++ caller (connect6.hpp + sample/cpp/main.cpp), compiled where it lies,
    links against libconnect6_amd.so with no unresolved symbol (it needs cpp_play, cpp_play_with, cpp_self_play)."""
    exe = tmp_path / "sample"
    libdir = os.path.join(ROOT, "alphazero-connect6_amd")
    cmd = ["g++", "-std=c++17", "-I/root/reference/Connect6/cppbind/headers", REF_SAMPLE, "-L" + libdir, "-lconnect6_amd",
           "-L/opt/rocm/lib", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)]
    subprocess.check_call(cmd)
    und = subprocess.check_output(["nm", "-u", str(exe)], text=True)
    assert "cpp_self_play" in und and "cpp_play" in und      # imported from the shared library, resolved at link time


def test_python_face_param_helpers(tmp_path):
    """default_param / param_to_tuple / dump_param(path, param=None) / load_param / board_size (pyconnect6/__init__.py:5-79)"""
    d = pkg.default_param()
    assert d == {"num_simulation": 800, "epsilon": 0.25, "dirichlet_alpha": 0.03, "c_puct": 1, "debug": False, "num_game_thread": 1}
    assert pkg.param_to_tuple(d) == (800, 0.25, 0.03, 1, False, 1)
    assert pkg.board_size() == 15
    base = str(tmp_path / "ckpt")
    pkg.dump_param(base)                                     # param=None -> the defaults
    assert os.path.exists(base + "_mcts.json") and pkg.load_param(base) == d
    d["num_simulation"] = 17
    pkg.dump_param(base, d)
    assert pkg.load_param(base) == d
    with pytest.raises(KeyError):
        pkg.param_to_tuple({"num_simulation": 2})
