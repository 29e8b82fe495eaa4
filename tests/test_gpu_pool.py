"""The per-game record pool against the reference's unbounded HashMap<u64, Node> (alphazero_policy/mod.rs:186-190,360-377):
growth is invisible in the results, exhaustion is never silent."""
import ctypes as C
import os

import numpy as np
import pytest

import c6oracle as o
from gpu_common import need_gpu, pkg

pytestmark = pytest.mark.gpu
N = 9


def peaked(player, boards, n=N):
    """A prior of e^(-14 d^2) around the centre-most empty cell (invariant under the 8 dihedral images, so the averaged prior
    keeps the peak; every other child is ~1e-6) and a value every mover likes two plies up (-0.9 * side to move: with
    two-stone turns the mover two plies above a leaf is always the other colour): with epsilon = 0.001 nearly every search
    follows the peaked child level after level -- the select depth reaches the teens (mean 6.5 on the oracle)"""
    b = np.asarray(boards).reshape(-1, n * n)
    rr, cc = np.divmod(np.arange(n * n), n)
    d2 = ((rr - n // 2) ** 2 + (cc - n // 2) ** 2).astype(np.float32)
    logits = np.where(b == 0, -14.0 * d2[None, :], -1e9).astype(np.float32)
    e = np.exp(logits - logits.max(axis=1, keepdims=True))
    pol = e / e.sum(axis=1, keepdims=True)
    return np.full(b.shape[0], -0.9 * player, np.float32), pol.astype(np.float32)


def oracle_eval(ctx, turn, nn, ln, boards, values, policies):
    b = np.ctypeslib.as_array(boards, shape=(ln, nn * nn))
    v, p = peaked(turn, b)
    np.ctypeslib.as_array(values, shape=(ln,))[:] = v
    np.ctypeslib.as_array(policies, shape=(ln, nn * nn))[:] = p
    return 0


def oracle_games(G, S, seed, eps):
    ref = o.self_play(N, num=G, threads=1, num_simulation=S, seed=seed, epsilon=eps, eval_kind=o.EVAL_CUSTOM, eval_fn=oracle_eval)
    return [([r * N + c for _, _, (r, c) in g["path"]], g["winner"]) for g in ref]


def stress_eval_fn(n):
    """the oracle's side of C6_EVAL_STRESS (every game peaked): pkg.stress_eval as an EVAL_CUSTOM callback"""
    def fn(ctx, turn, nn, ln, boards, values, policies):
        b = np.ctypeslib.as_array(boards, shape=(ln, nn * nn))
        v, p = pkg.stress_eval(nn, turn, b)
        np.ctypeslib.as_array(values, shape=(ln,))[:] = v
        np.ctypeslib.as_array(policies, shape=(ln, nn * nn))[:] = p
        return 0
    return fn


def test_trees_grow_past_the_per_game_caps_and_games_stay_exact():
    """A pool sized for 8 records per game (rounded up to one move's worth: 2 pages) and a prior that keeps several times
    num_simulation records alive: the games take page after page from the shared pool at their move boundaries, c6_engine_run
    raises the per-game caps when they reach them, and every game still equals the oracle's unbounded map stone for stone."""
    need_gpu()
    G, S, seed, eps = 4, 60, 5, 0.001
    eng = pkg.Engine(board_size=N, num_games=G, num_simulation=S, seed=seed, epsilon=eps, evaluator=pkg.EVAL_CALLBACK, records_per_game=8)
    eng.set_callback(peaked)
    st0 = eng.stats()
    assert st0["records_per_game"] == 4 * 64 and st0["pool_pages_free"] == st0["pool_pages"] - G * 1      # every game holds one move's worth (60 records: a page)
    eng.run()
    st = eng.stats()
    games = {g["uid"]: g for g in eng.fetch_games()}
    eng.close()
    assert st["overflow_games"] == 0 and st["abandoned_games"] == 0 and len(games) == G
    assert st["cap_raises"] >= 1 and st["records_per_game"] >= 8 * 64 and st["max_depth"] > 8 and st["max_records_used"] > 4 * 64, st
    assert st["pool_pages_free"] == st["pool_pages"]                                            # finished games hold nothing
    for uid, (mv, w) in enumerate(oracle_games(G, S, seed, eps)):
        assert list(games[uid]["moves"]) == mv and games[uid]["winner"] == w


def test_step_without_maintain_parks_the_game_and_maintain_resumes_it_exactly():
    """c6_engine_step never synchronises, so it cannot raise the caps: a game that needs more pages than the cap allows waits at its
    move boundary (visible in the stats: parked_games), loses nothing, and continues after c6_engine_maintain -- same final game as
    the oracle."""
    need_gpu()
    G, S, seed, eps = 2, 60, 9, 0.001
    eng = pkg.Engine(board_size=N, num_games=G, num_simulation=S, seed=seed, epsilon=eps, evaluator=pkg.EVAL_CALLBACK, records_per_game=8)
    eng.set_callback(peaked)
    sims = -1
    for _ in range(400):
        eng.step(S)
        st = eng.stats()
        if st["parked_games"] == G and st["simulations"] == sims:
            break
        sims = st["simulations"]
    assert st["parked_games"] == G and st["live_games"] == 0 and st["overflow_games"] == G, st
    assert st["simulations"] % S == 0                         # parked at move boundaries: no search was cut short
    eng.maintain()
    eng.step(1)
    st = eng.stats()
    assert st["parked_games"] == 0 and st["live_games"] == G and st["cap_raises"] == 1, st
    eng.run()
    games = {g["uid"]: g for g in eng.fetch_games()}
    eng.close()
    for uid, (mv, w) in enumerate(oracle_games(G, S, seed, eps)):
        assert list(games[uid]["moves"]) == mv and games[uid]["winner"] == w


def test_exhausted_pool_returns_partial_games_and_says_so():
    """growth capped (max_records_per_game): c6_engine_run finishes what it can, returns C6_ERR_OVERFLOW with a message and
    hands the stopped games back as partial results (winner 0, a prefix of the oracle's game) -- never C6_OK with games
    missing, never an abort."""
    need_gpu()
    G, S, seed, eps = 3, 40, 5, 0.001
    eng = pkg.Engine(board_size=N, num_games=G, num_simulation=S, seed=seed, epsilon=eps, evaluator=pkg.EVAL_CALLBACK, records_per_game=8,
                     max_records_per_game=128)
    eng.set_callback(peaked)
    with pytest.raises(pkg.C6Error) as ei:
        eng.run()
    assert "partial" in str(ei.value)
    st = eng.stats()
    games = {g["uid"]: g for g in eng.fetch_games()}
    eng.close()
    assert len(games) == G and st["abandoned_games"] >= 1 and st["overflow_games"] == st["abandoned_games"]
    ref = oracle_games(G, S, seed, eps)
    partial = 0
    for uid, (mv, w) in enumerate(ref):
        got = list(games[uid]["moves"])
        if len(got) < len(mv):
            partial += 1
            assert games[uid]["winner"] == 0 and got == mv[:len(got)]
        else:
            assert got == mv and games[uid]["winner"] == w
    assert partial == st["abandoned_games"]


def test_shared_pool_grows_by_mapping_and_pages_change_hands():
    """4096 games on the device-side stress evaluator (C6_EVAL_STRESS: every fourth game's tree keeps ~6 x num_simulation records
    alive, the others stay small), a pool created for one move's worth per game: the deep games take what the shallow ones never
    need, the pool maps more memory when it runs low (pool_grows), pages return when games end.  Sampled games of both kinds equal
    the oracle stone for stone; at the end every page is back in the pool."""
    need_gpu()
    n, G, S, seed, eps, every = 9, 4096, 60, 13, 0.001, 4
    eng = pkg.Engine(board_size=n, num_games=G, num_simulation=S, seed=seed, epsilon=eps, evaluator=pkg.EVAL_STRESS, stress_every=every,
                     records_per_game=8)
    st0 = eng.stats()
    eng.run()
    st = eng.stats()
    games = {g["uid"]: g for g in eng.fetch_games()}
    eng.close()
    assert len(games) == G and st["overflow_games"] == 0 and st["abandoned_games"] == 0
    assert st["pool_pages_free"] == st["pool_pages"] and st["pool_pages_low"] < st0["pool_pages_free"]
    assert st["pool_grows"] >= 1 and st["pool_pages"] > st0["pool_pages"] and st["pool_bytes"] > st0["pool_bytes"], (st0, st)
    assert st["max_records_used"] > 4 * S and st["cap_raises"] >= 1
    fn = stress_eval_fn(n)
    for uid in (0, 4, 1024, 4092):                              # peaked
        ref = o.self_play(n, num=1, threads=1, num_simulation=S, seed=seed, epsilon=eps, eval_kind=o.EVAL_CUSTOM, eval_fn=fn, uid_base=uid)[0]
        assert list(games[uid]["moves"]) == [r * n + c for _, _, (r, c) in ref["path"]] and games[uid]["winner"] == ref["winner"], uid
    for uid in (1, 2, 1023, 4095):                              # uniform
        ref = o.self_play(n, num=1, threads=1, num_simulation=S, seed=seed, epsilon=eps, eval_kind=o.EVAL_UNIFORM, uid_base=uid)[0]
        assert list(games[uid]["moves"]) == [r * n + c for _, _, (r, c) in ref["path"]] and games[uid]["winner"] == ref["winner"], uid


def test_page_churn_under_recycling_keeps_every_game_exact():
    """Steady-state mode at scale: 2048 staggered 9x9 games that restart as they finish (recycle), every third on the stress evaluator, a pool
    created for one move's worth per game -- pages and table chunks go back and forth between finishing shallow games and growing deep ones
    for ~30 game lengths while the host side raises the per-game caps as it goes (the pool's first mapping is a whole VMM unit: large enough here).  No game is abandoned or left parked, and sampled games of
    both kinds, from the second generation of their slots to the last, equal the oracle stone for stone."""
    need_gpu()
    n, G, S, seed, eps, every = 9, 2048, 40, 23, 0.001, 3
    eng = pkg.Engine(board_size=n, num_games=G, num_simulation=S, seed=seed, epsilon=eps, evaluator=pkg.EVAL_STRESS, stress_every=every,
                     records_per_game=8, recycle=True, stagger=True)
    st0 = eng.stats()
    done = {}
    for it in range(60):
        eng.step(S * 20)
        eng.maintain()
        for g in eng.fetch_games():
            done[g["uid"]] = g
    st = eng.stats()
    eng.close()
    assert st["abandoned_games"] == 0 and st["overflow_games"] == st["parked_games"] and st["parked_games"] <= G // 100, st
    assert st["cap_raises"] >= 1 and st["max_records_used"] > 3 * S and st["pool_pages_low"] < st0["pool_pages_free"] // 2, (st0, st)
    assert len(done) > 8 * G and max(done) >= 8 * G                       # many generations of every slot
    fn = stress_eval_fn(n)
    uids = sorted(done)
    # (a staggered slot's FIRST game starts its first move with uid % S simulations already counted -- connect6_amd.h: stagger -- so the
    #  oracle's game of that uid is another one; every later generation of the slot is a whole game)
    peaked = [u for u in uids if u % every == 0 and u >= G]
    plain = [u for u in uids if u % every != 0 and u >= G]
    for u in peaked[:2] + peaked[len(peaked) // 2:len(peaked) // 2 + 2] + peaked[-2:] + plain[:2] + plain[-2:]:
        kind = dict(eval_kind=o.EVAL_CUSTOM, eval_fn=fn) if u % every == 0 else dict(eval_kind=o.EVAL_UNIFORM)
        ref = o.self_play(n, num=1, threads=1, num_simulation=S, seed=seed, epsilon=eps, uid_base=u, **kind)[0]
        assert list(done[u]["moves"]) == [r * n + c for _, _, (r, c) in ref["path"]] and done[u]["winner"] == ref["winner"], u


def test_recycled_error_slots_stay_consistent_without_stagger():
    """recycle = 1, stagger = 0 and a host evaluator.  Games that cannot be served (max_records_per_game = 128) are abandoned
    and their slots restart OUT OF PHASE with the aligned move rounds; K1 runs every round, so every move of a restarted game is applied
    with its own winner / err: the canonical boards (K1's state) and the tree kernels' root positions stay identical for every slot, and
    every game that finishes is the oracle's game."""
    need_gpu()
    G, S, seed, eps = 4, 20, 17, 0.001
    eng = pkg.Engine(board_size=N, num_games=G, num_simulation=S, seed=seed, epsilon=eps, evaluator=pkg.EVAL_CALLBACK, records_per_game=8,
                     max_records_per_game=128, recycle=True)
    eng.set_callback(peaked)
    done = []
    for it in range(60):
        eng.step(37)                                            # not a multiple of S: calls end in the middle of moves
        eng.maintain()
        boards, turn, remain = eng.boards()
        for g in range(G):
            r = eng.root(g)
            if r["status"] == 0:
                assert (r["board"].reshape(-1) == boards[g]).all() and r["turn"] == turn[g] and r["remain"] == remain[g], (it, g)
        done += eng.fetch_games()
    st = eng.stats()
    eng.close()
    assert st["abandoned_games"] >= 1, st
    uids = [g["uid"] for g in done]
    assert len(uids) == len(set(uids))
    fn = None
    for g in done[:6]:
        ref = o.self_play(N, num=1, threads=1, num_simulation=S, seed=seed, epsilon=eps, eval_kind=o.EVAL_CUSTOM, eval_fn=oracle_eval, uid_base=g["uid"])[0]
        assert list(g["moves"]) == [r * N + c for _, _, (r, c) in ref["path"]] and g["winner"] == ref["winner"], g["uid"]


def test_dropin_self_play_survives_an_exhausted_pool(monkeypatch, capfd):
    """cpp_self_play with the pool capped by the environment knobs: no abort; every game comes back, the stopped ones as
    {winner 0, stones so far} with the reason on stderr (SURVEY 8b "Errors")."""
    need_gpu()
    monkeypatch.setenv("C6_RECORDS", "48")
    monkeypatch.setenv("C6_MAX_RECORDS", "128")      # two pages per game: the first moves fit, the trees the peaked prior grows do not
    monkeypatch.setenv("C6_SEED", "3")
    n = pkg.REF_N

    def pol(turn, boards):
        return peaked(turn, boards, n)

    param = pkg.default_param()
    param.update(num_simulation=40, epsilon=0.001, num_game_thread=2)
    res = pkg.self_play_np(pol, param)
    err = capfd.readouterr().err
    assert len(res) == 2 and "partial" in err
    for winner, path in res:
        assert winner == 0 and 1 <= len(path) < n * n


DRY_POOL_CHILD = r"""
import json, sys
sys.path.insert(0, sys.argv[2])
from c6amd_loader import load_package
pkg = load_package()
n, G, S, seed, eps, every = 9, 4096, 40, 29, 0.001, 2
eng = pkg.Engine(board_size=n, num_games=G, num_simulation=S, seed=seed, epsilon=eps, evaluator=pkg.EVAL_STRESS, stress_every=every,
                 records_per_game=8, recycle=True, stagger=True)
audits, done = [], {}
for it in range(48):
    eng.step(S * 6 + 7)
    eng.maintain()
    audits.append(eng.pool_audit())
    for g in eng.fetch_games():
        done[g["uid"]] = (int(g["winner"]), [int(m) for m in g["moves"]])
st = eng.stats()
eng.close()
keep = sorted(u for u in done if u >= G)
pick = keep[:3] + keep[len(keep) // 2:len(keep) // 2 + 3] + keep[-3:]
json.dump(dict(audits=audits, st=st, ndone=len(done), games={str(u): done[u] for u in pick}), open(sys.argv[1], "w"))
"""


def test_nearly_dry_pool_never_hands_a_page_to_two_games(tmp_path):
    """with the pool nearly dry many parked waves retry ring_pop at every round's head.  The pool here cannot grow
    (C6_NO_VMM=1: plain allocations of the initial two pages per game) while every second of 4096 recycling games grows a deep tree:
    games park, are abandoned and restarted, pages and chunks change hands all the time.  After every chunk of rounds
    c6_engine_pool_audit must find every page / chunk id exactly once -- held by one game or in its free ring: no id twice (two
    owners), none lost -- and the games that do finish are still the oracle's games."""
    need_gpu()
    import json
    import subprocess
    import sys
    out = str(tmp_path / "out.json")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run([sys.executable, "-c", DRY_POOL_CHILD, out, root], check=True, env=dict(os.environ, C6_NO_VMM="1"), timeout=600)
    d = json.load(open(out))
    st = d["st"]
    assert st["pool_grows"] == 0 and st["pool_pages_low"] <= 4 and st["abandoned_games"] >= 10, st          # the pool did run dry
    for k, a in enumerate(d["audits"]):
        assert a["pages_dup"] == a["pages_leaked"] == a["pages_bad"] == 0, (k, a)
        assert a["chunks_dup"] == a["chunks_leaked"] == a["chunks_bad"] == 0, (k, a)
        assert a["pages_held"] + a["pages_free"] == a["pages_total"] and a["chunks_held"] + a["chunks_free"] == a["chunks_total"], (k, a)
    assert d["ndone"] > 4096 and len(d["games"]) >= 6
    n, S, seed, eps, every = 9, 40, 29, 0.001, 2
    fn = stress_eval_fn(n)
    for u, (winner, moves) in d["games"].items():
        u = int(u)
        kind = dict(eval_kind=o.EVAL_CUSTOM, eval_fn=fn) if u % every == 0 else dict(eval_kind=o.EVAL_UNIFORM)
        ref = o.self_play(n, num=1, threads=1, num_simulation=S, seed=seed, epsilon=eps, uid_base=u, **kind)[0]
        assert moves == [r * n + c for _, _, (r, c) in ref["path"]] and winner == ref["winner"], u


def test_ring_pop_protocol_selftest():
    """The pop protocol on its own (c6_pool_ring_selftest): thousands of launches in which a few / many waves ask for 1-8 ids from a
    ring that holds far fewer -- almost all of every launch runs on a nearly dry ring, single-id requests succeeding beside larger
    ones failing.  No id is handed out twice, none beyond the ring's content, and nearly all are handed out (a failure in flight can
    only make another request fail).  Round 3's add-then-undo protocol hands out duplicates in the first three geometries (4352 /
    286 / 205 of them: profiles/r04_ring_selftest.txt, A/B build tools/ab_patches/tree_ring_pop_add_undo.sed)."""
    need_gpu()
    # `least`: a loose floor under the fewest ids one launch handed out (observed 13 / 9 / 17 / 993 / 30: which requests meet a momentarily
    # depressed credit count depends on the waves' timing) -- the invariants are dup == over == 0
    for geo, least in ((dict(ids=16, waves=4, tries=8, launches=30000, seed=3), 4), (dict(ids=16, waves=8, tries=6, launches=30000, seed=4), 4),
                       (dict(ids=24, waves=16, tries=4, launches=30000, seed=6), 8), (dict(ids=1000, waves=4096, tries=4, launches=200, seed=5), 900),
                       (dict(ids=37, waves=2048, tries=8, launches=1000, seed=9), 16)):
        r = pkg.pool_ring_selftest(**geo)
        assert r["dup"] == 0 and r["over"] == 0 and r["min_popped"] >= least, (geo, r)
