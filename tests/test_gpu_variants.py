"""The library's alternative paths:

* C6_BOARD_WIDE=1 (child process: the switch is read once per process) -- K1's global-load kernel, which the launcher also takes
  when a board array is too large for the 32-bit row descriptors of the default kernel (board_kernel.hip): same bits as the default
  kernel and as the oracle (game_impl/mod.rs:148-169, search_winner/mod.rs:201-246);
* the FP32 trainer's two implementations, both selectable through the API: C6_PREC_FP32 (Winograd-domain forward / input gradient /
  weight gradient, conv_wino4.hip) against C6_PREC_FP32_DIRECT (the direct kernels of conv_f32n.hip / train_kernels.hip) -- two
  independent implementations of the same sums at a batch large enough for 64-tile blocks and several chunks.

* C6_NO_VMM=1 (child process) -- the record pools as plain allocations of their initial size, what a device without HIP virtual memory
  management gets (engine.cpp: VmRange): the pool cannot grow, so deep trees end as partial games with the reason said -- never silently.

* C6_W6_SPLIT=0 / 1, C6_W4_SPLIT=0 / 1 (child processes) -- the two work decompositions of conv_wino6.hip's / conv_wino4.hip's GEMM kernel: a block of 64 tiles per workgroup (the
  benchmark's batches) or one ROW of a block's positions per workgroup (small batches: a block is a chain of up to 48 positions on one CU).
  The same instruction stream per position, so the net's outputs are equal BIT FOR BIT at every batch size.

(The forms that lost their A/B comparisons -- separate transform kernels, a stem kernel of its own, unfused heads, the two-workgroup
weight-gradient kernel -- are no longer in the product: compile-time options of A/B builds only, tools/ab_build.sh.)
"""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

import c6oracle as o
from gpu_common import need_gpu, pkg

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from c6amd_loader import load_package
pkg = load_package()
d = np.load(sys.argv[1])
out = {}
n = int(d["n"])
out["winner"] = pkg.search_winner(n, d["boards"])
b, t, r, w, e = pkg.board_step(n, d["boards"], d["turn"], d["remain"], d["pos"])
out.update(b=b, t=t, r=r, w=w, e=e, mask=pkg.legal_mask(n, d["boards"]))
np.savez(sys.argv[2], **out)
""" % (ROOT, os.path.join(ROOT, "tests"))


POOL_CHILD = r"""
import sys, json, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from c6amd_loader import load_package
pkg = load_package()
n, G, S = 9, 64, 60
eng = pkg.Engine(board_size=n, num_games=G, num_simulation=S, seed=3, epsilon=0.001, evaluator=pkg.EVAL_STRESS, records_per_game=8)
st0 = eng.stats()
err = ""
try:
    eng.run()
except pkg.C6Error as e:
    err = str(e)
st = eng.stats()
games = eng.fetch_games()
json.dump(dict(err=err, st0=st0, st=st, lens=[len(g["moves"]) for g in games], winners=[g["winner"] for g in games]), open(sys.argv[1], "w"))
""" % (ROOT, os.path.join(ROOT, "tests"))


def test_pools_without_vmm_cannot_grow_and_say_so():
    need_gpu()
    import json
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "out.json")
        subprocess.run([sys.executable, "-c", POOL_CHILD, out], check=True, env=dict(os.environ, C6_NO_VMM="1"), timeout=300)
        d = json.load(open(out))
    assert d["st0"]["pool_pages"] == 64 * 2 and d["st"]["pool_pages"] == 64 * 2 and d["st"]["pool_grows"] == 0      # 2 pages per game, for good
    assert "partial" in d["err"] and d["st"]["abandoned_games"] >= 1
    assert len(d["lens"]) == 64                                          # every game comes back: finished ones and partial ones
    assert sum(1 for w, l in zip(d["winners"], d["lens"]) if w == 0 and l < 81) == d["st"]["abandoned_games"]


def run_child(env_extra, **arrays):
    with tempfile.TemporaryDirectory() as td:
        fin, fout = os.path.join(td, "in.npz"), os.path.join(td, "out.npz")
        np.savez(fin, **arrays)
        env = dict(os.environ, **env_extra)
        subprocess.run([sys.executable, "-c", CHILD, fin, fout], check=True, env=env, timeout=300)
        with np.load(fout) as d:
            return {k: d[k] for k in d.files}


def test_board_global_load_kernel_matches(n):
    need_gpu()
    rng = np.random.default_rng(100 + n)
    G = 1027      # ragged
    boards = rng.choice(np.array([-1, 0, 1], np.int8), size=(G, n, n), p=[0.3, 0.4, 0.3])
    soa = pkg.to_soa(boards)
    turn = rng.choice(np.array([-1, 1], np.int8), G)
    remain = rng.integers(1, 3, G).astype(np.int8)
    pos = rng.integers(-1, n * n + 2, G).astype(np.int32)      # includes "no move" (-1) and invalid positions
    here = dict(winner=pkg.search_winner(n, soa), mask=pkg.legal_mask(n, soa))
    here["b"], here["t"], here["r"], here["w"], here["e"] = pkg.board_step(n, soa, turn, remain, pos)
    other = run_child({"C6_BOARD_WIDE": "1"}, n=n, boards=soa, turn=turn, remain=remain, pos=pos)
    for k, v in here.items():
        assert (other[k] == v).all(), k
    assert (here["winner"] == np.array([o.search(b) for b in boards], np.int8)).all()


def test_winograd_weight_gradient_matches_direct_kernel():
    """The FP32 trainer's two weight-gradient implementations on the SAME dy and x (c6_conv_wgrad_op): in the Winograd domain (dU = sum
    over tiles of A dY A^T x B^T x B, dg = G^T dU G; c6_conv_wino4_wgrad) against the direct kernel (c6_wgrad_kernel) -- 15x15 (mixed
    tiling) and 19x19, 300 boards: every tile region holds whole and partial blocks and the reduction is cut into chunks.  Same sums,
    different order and different intermediate values: within 1e-5 in L2 of each other, and both within 1e-5 of float64 on a slice."""
    need_gpu()
    for n in (15, 19):
        C, B = 256, 300
        rng = np.random.default_rng(7 + n)
        x = np.maximum(rng.standard_normal((B, n * n, C)), 0).astype(np.float32)          # post-ReLU activations
        dy = (rng.standard_normal((B, n * n, C)) * 0.1).astype(np.float32)
        direct = pkg.conv_wgrad(n, C, dy, x, 0)
        wino = pkg.conv_wgrad(n, C, dy, x, 1)
        assert np.linalg.norm(wino - direct) <= 1e-5 * np.linalg.norm(direct), (n, np.linalg.norm(wino - direct) / np.linalg.norm(direct))
        # float64 reference of dW[co, ci, a, b] = sum_{board, y, x} dy[., y, x, co] * x[., y + a - 1, x + b - 1, ci] for 8 x 8 channel pairs
        xs = np.zeros((B, n + 2, n + 2, 8), np.float64)
        xs[:, 1:-1, 1:-1] = x.reshape(B, n, n, C)[..., 5:13]
        dys = dy.reshape(B, n, n, C)[..., 100:108].astype(np.float64)
        want = np.stack([np.stack([np.einsum("byxo,byxi->oi", dys, xs[:, a:a + n, b:b + n]) for b in range(3)], -1) for a in range(3)], -2)
        for got in (direct, wino):
            g = got[100:108, 5:13].astype(np.float64)
            assert np.linalg.norm(g - want) <= 1e-5 * np.linalg.norm(want), n


def test_trainer_fp32_paths_agree_on_a_step():
    """a whole FP32 training step on the Winograd-domain kernels (C6_PREC_FP32) and on the direct kernels (C6_PREC_FP32_DIRECT): the same
    loss to 1e-6; the gradients agree as far as ReLU allows -- the two forward passes round differently (~1e-6), a few ReLU inputs next
    to zero fall on the other side, and the gradient is discontinuous there (tests/test_gpu_train.py explains the protocol): L2 5e-3."""
    need_gpu()
    import __graft_entry__ as ge
    n, C, blocks, B = 15, 256, 2, 300
    rng = np.random.default_rng(22)
    blob = ge.make_random_blob(n, C, blocks, rng)
    batch = (rng.integers(-1, 2, B).astype(np.float32), rng.choice([-1.0, 1.0], B).astype(np.float32),
             rng.integers(-1, 2, (B, n * n)).astype(np.float32), rng.integers(0, n * n, B).astype(np.float32))
    res = []
    for prec in (pkg.PREC_FP32, pkg.PREC_FP32_DIRECT):
        t = pkg.Trainer(n, C, blocks, B, lr=1e-3, momentum=0.9, precision=prec)
        t.load_weights(blob)
        res.append((float(t.step(*batch)), t.gradients()))
        t.close()
    (la, ga), (lb, gb) = res
    assert abs(la - lb) <= 1e-6 * abs(lb)
    assert np.linalg.norm(ga - gb) <= 5e-3 * np.linalg.norm(gb), np.linalg.norm(ga - gb) / np.linalg.norm(gb)


def test_trainer_default_precision_is_the_engines():
    """C6_PREC_DEFAULT resolves to the same arithmetic in the engine and in the trainer: f32 (model.hpp:52-108 is an f32 net and
    main.cpp:107-157 trains it in f32) -- a default-precision step equals an explicit C6_PREC_FP32 step bit for bit."""
    need_gpu()
    import __graft_entry__ as ge
    n, C, blocks, B = 9, 64, 2, 16
    rng = np.random.default_rng(3)
    blob = ge.make_random_blob(n, C, blocks, rng)
    batch = (rng.integers(-1, 2, B).astype(np.float32), rng.choice([-1.0, 1.0], B).astype(np.float32),
             rng.integers(-1, 2, (B, n * n)).astype(np.float32), rng.integers(0, n * n, B).astype(np.float32))
    out = []
    for prec in (pkg.PREC_DEFAULT, pkg.PREC_FP32):
        t = pkg.Trainer(n, C, blocks, B, lr=1e-3, momentum=0.9, precision=prec)
        t.load_weights(blob)
        out.append((float(t.step(*batch)), t.gradients(), t.export_weights()))
        t.close()
    assert out[0][0] == out[1][0] and (out[0][1] == out[1][1]).all() and (out[0][2] == out[1][2]).all()


W4_CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from c6amd_loader import load_package
import __graft_entry__ as ge
pkg = load_package()
out = {}
for n, blocks in ((15, 3), (9, 4), (19, 2)):      # (19 x 19 with one residual block: conv_wino4.hip's tiling, conv_wino6.hip needs two)
    blob = ge.make_random_blob(n, 256, blocks, np.random.default_rng(6))
    for num in (1, 13, 300, 2000):
        boards = np.random.default_rng(num).integers(-1, 2, size=(num, n * n)).astype(np.int8)
        pol, val = pkg.net_forward(n, 256, blocks, blob, boards, 1 if num %% 2 else -1, precision=pkg.PREC_FP32)
        out["p%%d_%%d" %% (n, num)] = pol
        out["v%%d_%%d" %% (n, num)] = val
np.savez(sys.argv[1], **out)
""" % (ROOT, os.path.join(ROOT, "tests"))


def test_w4_gemm_rows_as_work_items_equal_blocks_bit_for_bit():
    """conv_wino4.hip: c6_wino4_gemm_kernel<N, true> (a row of positions per workgroup; and its four-wave half-width jobs) against <N, false> (a block per workgroup) through
    c6_net_forward at 15 x 15, 9 x 9 and (two-block tower) 19 x 19, 1 .. 2000 boards: equal bit for bit."""
    need_gpu()
    res = {}
    with tempfile.TemporaryDirectory() as td:
        for form, env in (("block", dict(C6_W4_SPLIT="0")), ("split", dict(C6_W4_SPLIT="1", C6_W4_HALF="0")), ("half", dict(C6_W4_SPLIT="1", C6_W4_HALF="1"))):
            out = os.path.join(td, f"w4_{form}.npz")
            # (C6_W6_N15=0: since r6 a 15 x 15 tower with >= 2 residual blocks runs conv_wino6.hip's tiling; the hook keeps it on the kernels under test)
            subprocess.run([sys.executable, "-c", W4_CHILD, out], check=True, env=dict(os.environ, C6_W6_N15="0", **env), timeout=600)
            res[form] = dict(np.load(out))
    assert sorted(res["block"]) == sorted(res["split"]) == sorted(res["half"]) and len(res["block"]) == 24
    for k in res["block"]:
        assert np.isfinite(res["block"][k]).all()
        assert np.array_equal(res["block"][k], res["split"][k]), k
        assert np.array_equal(res["block"][k], res["half"][k]), k


W6_CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from c6amd_loader import load_package
import __graft_entry__ as ge
pkg = load_package()
out = {}
for n, blocks in ((19, 4), (19, 8), (15, 10)):      # (r6: the reference's own shape, 15 x 15 with stem + 9 blocks, runs the same GEMM kernels over its own tiling)
    blob = ge.make_random_blob(n, 256, blocks, np.random.default_rng(5))
    # (10 000 boards, stem + 3 blocks only: 313 blocks of class D' > the 256 workgroups of a grid row -- the forced split form's block loop makes a SECOND
    #  trip in the five-position row form too, ADVICE r5)
    for num in (1, 13, 40, 333, 1100) + ((10000,) if blocks == 4 else ()):
        boards = np.random.default_rng(num).integers(-1, 2, size=(num, n * n)).astype(np.int8)
        pol, val = pkg.net_forward(n, 256, blocks, blob, boards, 1 if num %% 2 else -1, precision=pkg.PREC_FP32)
        out["p%%d_%%d_%%d" %% (n, blocks, num)] = pol
        out["v%%d_%%d_%%d" %% (n, blocks, num)] = val
np.savez(sys.argv[1], **out)
""" % (ROOT, os.path.join(ROOT, "tests"))


def test_w6_gemm_rows_as_work_items_equal_blocks_bit_for_bit():
    """conv_wino6.hip: c6_w6_gemm_kernel<., true> (r5: a row of positions per workgroup -- one game's boards on 28 CUs instead of 4; and <., true, true>, the
    four-wave half-width jobs of the tiniest batches, forced at every size) against
    <., false> (a block per workgroup) through c6_net_forward at 1 .. 1100 boards (1 board: every region's block nearly empty; 333: full and
    partial blocks; 1100 boards: 344 + 35 blocks -- the block loop of both forms runs more than once; 10 000 boards: 313 blocks of class D', more than a
    grid row of workgroups, so the five-position split jobs make a second trip too), stem + 3 and stem + 7 blocks."""
    need_gpu()
    res = {}
    with tempfile.TemporaryDirectory() as td:
        # block form / split form with eight-wave jobs / split form with the four-wave half-width jobs of the tiniest batches (forced at every size)
        # (split / half: both row forms in ONE launch, c6_w6_gemm_split_pair_kernel / c6_w6_gemm_half_pair_kernel; split2: the split form as the two launches it was)
        for form, env in (("block", dict(C6_W6_SPLIT="0")), ("split", dict(C6_W6_SPLIT="1", C6_W6_HALF="0")), ("half", dict(C6_W6_SPLIT="1", C6_W6_HALF="1")),
                          ("split2", dict(C6_W6_SPLIT="1", C6_W6_HALF="0", C6_W6_PAIR="0"))):
            out = os.path.join(td, f"w6_{form}.npz")
            subprocess.run([sys.executable, "-c", W6_CHILD, out], check=True, env=dict(os.environ, **env), timeout=600)
            res[form] = dict(np.load(out))
    assert sorted(res["block"]) == sorted(res["split"]) == sorted(res["half"]) == sorted(res["split2"]) and len(res["block"]) == 32
    for k in res["block"]:
        assert np.isfinite(res["block"][k]).all()
        assert np.array_equal(res["block"][k], res["split"][k]), k
        assert np.array_equal(res["block"][k], res["half"][k]), k
        assert np.array_equal(res["block"][k], res["split2"][k]), k


N15_CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from c6amd_loader import load_package
import __graft_entry__ as ge
pkg = load_package()
out = {}
for blocks in (3, 10):
    blob = ge.make_random_blob(15, 256, blocks, np.random.default_rng(8))
    for num in (5, 333):
        boards = np.random.default_rng(num).integers(-1, 2, size=(num, 225)).astype(np.int8)
        pol, val = pkg.net_forward(15, 256, blocks, blob, boards, 1 if num %% 2 else -1, precision=pkg.PREC_FP32)
        out["p%%d_%%d" %% (blocks, num)] = pol
        out["v%%d_%%d" %% (blocks, num)] = val
np.savez(sys.argv[1], **out)
""" % (ROOT, os.path.join(ROOT, "tests"))


def test_15x15_tilings_agree_and_match_the_oracle():
    """r6: the 15 x 15 tower (the reference's board: Connect6/libconnect6/src/lib.rs:21) on conv_wino6.hip's tiling -- 6 + 6 + 3 across, 4 + 4 + 4 + 3 down,
    488 positions per board -- against conv_wino4.hip's F(4x4) tiling of the same board (534 positions; C6_W6_N15=0) and against the CPU oracle's direct
    convolution: two different Winograd decompositions and a direct sum of the same f32 net, all within 1e-5 (north_star's tolerance), stem + 2 and the
    reference's stem + 9 blocks, a handful of boards and full + partial blocks of every tile class."""
    need_gpu()
    import __graft_entry__ as ge
    import c6oracle as o
    res = {}
    with tempfile.TemporaryDirectory() as td:
        for form, env in (("w6", {}), ("w4", dict(C6_W6_N15="0"))):
            out = os.path.join(td, f"n15_{form}.npz")
            subprocess.run([sys.executable, "-c", N15_CHILD, out], check=True, env=dict(os.environ, **env), timeout=600)
            res[form] = dict(np.load(out))
    assert sorted(res["w6"]) == sorted(res["w4"]) and len(res["w6"]) == 8
    differ = 0
    for k in res["w6"]:
        assert np.isfinite(res["w6"][k]).all()
        assert np.abs(res["w6"][k] - res["w4"][k]).max() <= 1e-5, k
        differ += not np.array_equal(res["w6"][k], res["w4"][k])
    assert differ > 0      # (two tilings round differently: identical bits would mean the hook did not switch anything)
    for blocks in (3, 10):
        net = o.Net(15, 256, blocks, ge.make_random_blob(15, 256, blocks, np.random.default_rng(8)))
        boards = np.random.default_rng(5).integers(-1, 2, size=(5, 225)).astype(np.int8)
        rpol, rval = net.forward(1.0, boards.astype(np.float32))
        assert np.abs(res["w6"][f"p{blocks}_5"] - rpol).max() <= 1e-5 and np.abs(res["w6"][f"v{blocks}_5"] - rval).max() <= 1e-5, blocks
        net.close()
