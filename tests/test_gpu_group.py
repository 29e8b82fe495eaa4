"""Several GPUs BEHIND the C ABI (include/connect6_amd.h part 6, csrc/group.cpp): the library shards a call's games over
devices itself -- one engine + host thread per device, uids from the same plan as shard.uid_plan -- and pools the training tuples natively
(ncclAllGather; host-staged when members share a device).  The reference does the same over cores inside the library: AsyncAgent::run
(agent/async_agent/mod.rs:100-137) behind cpp_self_play (cppbind/libcppconnect6/src/lib.rs:110-122), results in any order.

On the one-GPU box: devices = [0, 0] / [0, 0, 0] are members sharing the GPU (the exchange goes through host memory -- RCCL admits one
rank per device) and devices = [0] is a one-member group whose pooling runs on the REAL RCCL calls (dlopen, ncclCommInitAll, ncclAllGather
on the side stream).  Bars: the same uid set, the same games stone for stone, the same tuple multiset as ONE engine playing all the games."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from gpu_common import need_gpu, pkg

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one_engine(n, G, S, seed, evaluator=None, **kw):
    eng = pkg.Engine(board_size=n, num_games=G, num_simulation=S, seed=seed, evaluator=evaluator or pkg.EVAL_RANDOM, **kw)
    return eng


def run_one_engine(n, G, S, seed, evaluator=None, blob=None, **kw):
    eng = one_engine(n, G, S, seed, evaluator, **kw)
    if blob is not None:
        eng.load_weights(blob)
    eng.enable_tuples(G * n * n)
    eng.run()
    games = {g["uid"]: g for g in eng.fetch_games()}
    import ctypes as C
    stride = pkg.lib().c6_tuple_stride(n)
    buf = np.zeros((G * n * n, stride), np.uint8)
    cnt = C.c_size_t()
    assert pkg.lib().c6_engine_fetch_tuples(eng.h, G * n * n, C.byref(cnt), buf.ctypes.data_as(C.c_void_p)) == 0
    eng.close()
    return games, sorted(bytes(r) for r in buf[: cnt.value])


def check_group_equals_engine(devices, n, G, S, seed, evaluator=None, blob=None, expect_rccl=None, **kw):
    grp = pkg.Group(devices=devices, board_size=n, num_games=G, num_simulation=S, seed=seed, evaluator=evaluator or pkg.EVAL_RANDOM, **kw)
    assert grp.size == len(devices)
    if blob is not None:
        grp.load_weights(blob)
    # the shard plan: a partition of [0, G) in member order, the first G % members members one game longer (shard.shard_games)
    first = 0
    for r in range(grp.size):
        mi = grp.member_info(r)
        assert mi["device"] == devices[r] and mi["first_uid"] == first and mi["games"] == G // grp.size + (1 if r < G % grp.size else 0)
        first += mi["games"]
    assert first == G
    assert not grp.uses_rccl                                                      # (the communicators are made when tuples are enabled)
    grp.enable_tuples(G * n * n)
    if expect_rccl is not None:
        assert grp.uses_rccl == expect_rccl
    grp.run()
    st = grp.stats()
    assert st["games_finished"] == G and st["live_games"] == 0 and st["overflow_games"] == 0
    got = {g["uid"]: g for g in grp.fetch_games()}
    pool = grp.pool_tuples()
    assert pool["dropped"] == 0 and len(pool["counts"]) == grp.size
    copies = [grp.fetch_pooled(r) for r in range(grp.size)]
    grp.close()
    want_games, want_tuples = run_one_engine(n, G, S, seed, evaluator, blob, **kw)
    assert sorted(got) == sorted(want_games) == list(range(G))                     # the same uid set
    for uid in range(G):                                                          # the same games, stone for stone
        assert list(got[uid]["moves"]) == list(want_games[uid]["moves"]) and got[uid]["winner"] == want_games[uid]["winner"], uid
    assert int(pool["counts"].sum()) == len(want_tuples) == st["tuples"]
    for r, c in enumerate(copies):                                                # EVERY member holds the whole pooled set
        assert sorted(bytes(x) for x in c) == want_tuples, r
    return pool


def test_two_members_on_one_device_equal_one_engine():
    """devices = {0, 0}: two engines on one GPU, driven by two host threads inside the library, pooled through host memory"""
    need_gpu()
    pool = check_group_equals_engine([0, 0], 9, 12, 12, 77, expect_rccl=False)
    assert (pool["counts"] > 0).all()


def test_three_members_unequal_shards_with_the_resident_net():
    """7 games over 3 members (3 + 2 + 2) with the resident net as the evaluator: weights replicated by c6_group_load_weights; a board's net output
    does not depend on the batch it travels in, so the sharded games equal the single engine's stone for stone"""
    need_gpu()
    import __graft_entry__ as ge
    n, C, blocks = 9, 64, 2
    blob = ge.make_random_blob(n, C, blocks, np.random.default_rng(5))
    check_group_equals_engine([0, 0, 0], n, 7, 8, 31, evaluator=pkg.EVAL_NET, blob=blob, net_channels=C, net_blocks=blocks, expect_rccl=False)


def test_one_member_pools_through_rccl():
    """devices = {0}: all devices distinct, so the exchange runs on librccl -- dlopen, ncclCommInitAll (one communicator), ncclAllGather on the member's
    side stream inside ncclGroupStart / End.  What one GPU can execute of the native RCCL path; the result must be the engine's own tuples."""
    need_gpu()
    check_group_equals_engine([0], 9, 10, 10, 5, expect_rccl=True)


def test_pipelined_exchange_keeps_every_record():
    """_begin / step / _end with recycling slots: the buffer an interval wrote is pooled while the next interval's rounds are enqueued behind the swap.
    Every record emitted (stats.tuples) is pooled exactly once -- none lost at a swap, none pooled twice -- and the pooled states replay legally.
    Both forms: host-staged (two members on one device) and RCCL (one member)."""
    need_gpu()
    n, G, S = 9, 64, 6
    for devices in ([0, 0], [0]):
        grp = pkg.Group(devices=devices, board_size=n, num_games=G, num_simulation=S, seed=9, evaluator=pkg.EVAL_UNIFORM, recycle=True, stagger=True)
        grp.enable_tuples(4096)
        pooled, intervals = [], 0
        grp.step(40)
        for _ in range(12):
            grp.pool_begin()
            grp.step(40)
            p = grp.pool_end()
            assert p["dropped"] == 0
            if p["padded"]:
                pooled.append(grp.fetch_pooled(0))
                intervals += 1
                assert (grp.fetch_pooled(grp.size - 1) == pooled[-1]).all()
        p = grp.pool_tuples()
        if p["padded"]:
            pooled.append(grp.fetch_pooled(0))
        st = grp.stats()
        finished = grp.fetch_games()
        grp.close()
        rec = np.concatenate(pooled, axis=0)
        assert intervals >= 3 and rec.shape[0] == st["tuples"] == sum(len(g["moves"]) for g in finished)
        assert len({bytes(r) for r in rec}) >= 0.5 * rec.shape[0]                 # (openings repeat across games; whole buffers must not)
        tup = pkg.decode_tuples(rec, n)
        # every pooled state is a prefix of a finished game with that game's winner as z
        want = {}
        for g in finished:
            for turn, board, cell in pkg.replay_moves(n, g["moves"]):
                want.setdefault(board.tobytes(), set()).add((turn, g["winner"], cell))
        for i in range(0, rec.shape[0], 5):
            assert (int(tup["turn"][i]), int(tup["z"][i]), int(tup["pos"][i])) in want[tup["state"][i].tobytes()], i


def test_group_callback_evaluator_serialised():
    """C6_EVAL_CALLBACK through a group: the members' host threads call one Python callable (serialised by the group); games equal ONE engine with the
    same callable"""
    need_gpu()
    n, G, S, seed = 9, 6, 6, 3

    def ev(player, boards):
        b = boards.astype(np.float32)
        # deterministic per board: a prior that prefers cells next to stones of the side to move, value from the stone balance
        nb = np.abs(np.roll(b.reshape(-1, n, n), 1, axis=2)).reshape(len(b), -1)
        return np.tanh(b.sum(axis=1) * 0.125 * player), (1.0 + nb) / (n * n)

    grp = pkg.Group(devices=[0, 0], board_size=n, num_games=G, num_simulation=S, seed=seed, evaluator=pkg.EVAL_CALLBACK)
    grp.set_callback(ev)
    grp.run()
    got = {g["uid"]: g for g in grp.fetch_games()}
    grp.close()
    eng = pkg.Engine(board_size=n, num_games=G, num_simulation=S, seed=seed, evaluator=pkg.EVAL_CALLBACK)
    eng.set_callback(ev)
    eng.run()
    want = {g["uid"]: g for g in eng.fetch_games()}
    eng.close()
    assert sorted(got) == sorted(want) == list(range(G))
    for uid in range(G):
        assert list(got[uid]["moves"]) == list(want[uid]["moves"]) and got[uid]["winner"] == want[uid]["winner"], uid


def test_cpp_self_play_shards_over_devices(monkeypatch):
    """the DROP-IN symbol itself: cpp_self_play with C6_DEVICES=0,0 plays its num_game_thread games on a two-member group and hands back the same
    games as the one-engine call (same C6_SEED, a deterministic callback) -- AlphaZero/cpp_weighted/main.cpp:143's call, unchanged, on every GPU"""
    need_gpu()
    n = pkg.REF_N

    def policy(turn, boards):
        b = np.asarray(boards, np.float32)
        return [float(np.tanh(x.sum() * 0.1)) for x in b], (1.0 / (n * n) + 0.001 * (b == 0)).tolist()

    param = dict(pkg.default_param(), num_simulation=4, num_game_thread=6)
    monkeypatch.setenv("C6_SEED", "12345")
    monkeypatch.delenv("C6_DEVICES", raising=False)
    one = pkg.self_play(policy, param)
    monkeypatch.setenv("C6_DEVICES", "0,0")
    two = pkg.self_play(policy, param)
    assert len(one) == len(two) == 6

    def key(res):
        winner, path = res
        return (winner, tuple((t, tuple(pos)) for t, _, pos in path))
    assert sorted(key(r) for r in one) == sorted(key(r) for r in two)
    for winner, path in two:                                                      # Path records: board BEFORE the move, replayed on the host
        stones = [r * n + c for _, _, (r, c) in path]
        for (turn, board, pos), (t2, b2, cell) in zip(path, pkg.replay_moves(n, stones)):
            assert turn == t2 and list(board) == b2.reshape(-1).tolist()


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_compiled_cpp_host_drives_a_group(tmp_path):
    """tools/cpp/group_selfplay.cpp: a C++ program with no Python and no launcher creates the group, plays to the end, pools the tuples and prints
    order-independent digests; the same digests come out of ONE engine here.  Run twice: members {0, 0} (host-staged) and {0} (RCCL)."""
    need_gpu()
    exe = tmp_path / "group_selfplay"
    libdir = os.path.join(ROOT, "alphazero-connect6_amd")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "cpp", "group_selfplay.cpp"),
                           "-L" + libdir, "-lconnect6_amd", "-Wl,-rpath," + libdir, "-o", str(exe)])
    n, G, S, seed = 9, 16, 10, 2024
    games, tuples = run_one_engine(n, G, S, seed)

    def fnv(rec):
        h = 1469598103934665603
        for b in rec:
            h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        return h
    hs = [fnv(t) for t in tuples]
    dsum, dxor = sum(hs) & 0xFFFFFFFFFFFFFFFF, 0
    for h in hs:
        dxor ^= h
    for devices, rccl in (("0,0", 0), ("0", 1)):
        r = json.loads(subprocess.check_output([str(exe), devices, str(n), str(G), str(S), str(seed), "random"], text=True, timeout=300).strip().splitlines()[-1])
        assert r["members"] == len(devices.split(",")) and r["uses_rccl"] == rccl, r
        assert r["games_finished"] == r["fetched_games"] == G and r["uid_sum"] == G * (G - 1) // 2 and r["overflow_games"] == 0, r
        assert r["len_sum"] == sum(len(g["moves"]) for g in games.values()) == r["pooled_records"] == len(tuples), r
        assert r["digest_sum"] == dsum and r["digest_xor"] == dxor and r["copies_equal"] == 1 and r["dropped"] == 0, r
    # steady state with the exchange overlapped (recycle + stagger, pooled every 32 rounds): nothing dropped, every emitted tuple pooled
    r = json.loads(subprocess.check_output([str(exe), "0,0", str(n), "64", "6", "7", "uniform", "512", "32"], text=True, timeout=300).strip().splitlines()[-1])
    assert r["rounds"] == 512 and r["dropped"] == 0 and r["pooled_records"] == r["tuples"] > 0 and r["payload_pools"] >= 3 and r["copies_equal"] == 1, r


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_compiled_cpp_host_runs_the_reference_outer_loop(tmp_path):
    """tools/cpp/group_loop.cpp: the reference trainer's outer loop (AlphaZero/cpp_weighted/main.cpp:138-168) as a C++ host over a group -- self-play
    sharded over the members with the resident net, tuples pooled, member 0's copy pushed device to device into the replay buffer, SGD steps sampled
    from it, the new weights loaded into the next epoch's group.  Two epochs on members {0, 0}: every stone played becomes exactly one pooled tuple and
    one buffer record, the losses are finite numbers, nothing was dropped."""
    need_gpu()
    exe = tmp_path / "group_loop"
    libdir = os.path.join(ROOT, "alphazero-connect6_amd")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "cpp", "group_loop.cpp"),
                           "-L" + libdir, "-lconnect6_amd", "-Wl,-rpath," + libdir, "-o", str(exe)])
    out = subprocess.check_output([str(exe), "0,0", "9", "64", "3", "48", "8", "2", "6", "128"], text=True, timeout=600)
    lines = [json.loads(l) for l in out.strip().splitlines() if l.startswith("{")]
    assert [l["epoch"] for l in lines] == [0, 1]
    total = 0
    for l in lines:
        assert l["members"] == 2 and l["games_finished"] == 48 and l["dropped"] == 0
        assert l["pooled"] == l["tuples"] == l["stones"] > 48 * 11
        total += l["stones"]
        assert l["buffer"] == total
        assert np.isfinite(l["loss_first"]) and np.isfinite(l["loss_last"]) and 0 < l["loss_last"] < 20


def test_group_argument_errors_are_reported_not_crashed():
    """the group's entry points refuse what they cannot do with a status and a message (c6_last_error), never with a fault: a device that is not there,
    fewer games than members, pooling before tuples are enabled, an _end without its _begin, tuples enabled twice; and a fetch with room for fewer
    records than were pooled hands out the first `cap` of them"""
    need_gpu()
    import ctypes as C
    L = pkg.lib()
    with pytest.raises(pkg.C6Error, match="not visible"):
        pkg.Group(devices=[0, 99], board_size=9, num_games=4)
    with pytest.raises(pkg.C6Error, match="fewer games than members"):
        pkg.Group(devices=[0, 0, 0], board_size=9, num_games=2)
    grp = pkg.Group(devices=[0, 0], board_size=9, num_games=6, num_simulation=4, seed=1, evaluator=pkg.EVAL_UNIFORM)
    with pytest.raises(pkg.C6Error, match="enable_tuples first"):
        grp.pool_begin()
    with pytest.raises(pkg.C6Error, match="without c6_group_pool_begin"):
        grp.pool_end()
    grp.enable_tuples(1000)
    with pytest.raises(pkg.C6Error, match="already enabled"):
        grp.enable_tuples(1000)
    grp.pool_begin()
    with pytest.raises(pkg.C6Error, match="has not been ended"):
        grp.pool_begin()
    p = grp.pool_end()
    assert p["padded"] == 0 and int(p["counts"].sum()) == 0 and grp.fetch_pooled(0).shape[0] == 0      # nothing finished yet: nothing travels
    assert L.c6_group_engine(grp.h, 2) is None and L.c6_group_member_info(grp.h, 5, None, None, None) != 0
    grp.run()
    p = grp.pool_tuples()
    total = int(p["counts"].sum())
    assert total == grp.stats()["tuples"] > 0
    part = grp.fetch_pooled(1, cap=7)
    assert part.shape[0] == 7 and (part == grp.fetch_pooled(1)[:7]).all()
    grp.close()
    grp.close()                                                                 # (idempotent)


def test_four_members_with_the_benchmark_net_equal_one_engine():
    """four members sharing the GPU (four host threads, four engines with the benchmark's kernels: 19x19, C = 256, stem + 3 blocks, S = 400), 4 x 512 games,
    6 rounds, against ONE engine with all 2048 slots: the same simulations, and slot i of member r has bit for bit the root of slot first_r + i -- a
    board's net output does not depend on the batch it travels in, nor on which of the GEMM's work decompositions the batch size selects."""
    need_gpu()
    import __graft_entry__ as ge
    n, C, blocks, G, S, R, seed = 19, 256, 4, 2048, 400, 6, 11
    blob = ge.make_random_blob(n, C, blocks, np.random.default_rng(4))
    grp = pkg.Group(devices=[0, 0, 0, 0], board_size=n, num_games=G, num_simulation=S, seed=seed, evaluator=pkg.EVAL_NET, net_channels=C, net_blocks=blocks)
    grp.load_weights(blob)
    grp.step(R)
    grp.sync()
    st = grp.stats()
    assert st["simulations"] == G * R and st["live_games"] == G and st["overflow_games"] == 0
    eng = pkg.Engine(board_size=n, num_games=G, num_simulation=S, seed=seed, evaluator=pkg.EVAL_NET, net_channels=C, net_blocks=blocks)
    eng.load_weights(blob)
    eng.step(R)
    for member in range(4):
        mi = grp.member_info(member)
        assert mi["games"] == 512 and mi["first_uid"] == 512 * member
        for slot in (0, 255, 511):
            a, b = grp.engine_root(member, slot), eng.root(mi["first_uid"] + slot)
            assert a["root_visit"] == b["root_visit"] == R and (a["visits"] == b["visits"]).all(), (member, slot)
            assert (a["q"] == b["q"]).all() and (a["prior"] == b["prior"]).all() and a["root_value"] == b["root_value"], (member, slot)
    eng.close()
    grp.close()
