"""The env-switched fallback paths of the library, each in a child process (the switches are read once per process):

* C6_BOARD_WIDE=1 -- K1's global-load kernel, which the launcher also takes when a board array is too large for the
  32-bit row descriptors of the default kernel (board_kernel.hip): same bits as the default kernel and as the oracle
  (game_impl/mod.rs:148-169, search_winner/mod.rs:201-246);
* C6_HEAD_FUSE=0 -- the evaluator with a separate 1x1 head convolution instead of the partial sums the tower's last
  convolution leaves behind (net_kernels.hip): same outputs within float summation order (model.hpp:80-92);
* C6_WINO_MID=0 -- the FP32 Winograd tower without the fused boundary kernel (conv_wino4.hip);
* C6_WINO_STEM=0 -- the same tower with the stem as a kernel of its own (its activations written and read back) instead of being
  computed inside the first input transform and the first block's boundary kernel;
* C6_WGRAD_WINO=0 -- the FP32 trainer's weight gradient by the direct kernel (train_kernels.hip) instead of the Winograd-domain one
  (conv_wino4.hip): two independent implementations of the same sums at a batch large enough for 64-tile blocks and several chunks.
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
if "train" in d:
    t = pkg.Trainer(int(d["n"]), int(d["C"]), int(d["blocks"]), int(d["B"]), lr=1e-3, momentum=0.9, precision=pkg.PREC_FP32)
    t.load_weights(d["blob"])
    out["loss"] = np.float64(t.step(d["w"], d["p"], d["b"], d["q"]))
    out["grads"] = t.gradients()
elif "boards" in d:
    n = int(d["n"])
    out["winner"] = pkg.search_winner(n, d["boards"])
    b, t, r, w, e = pkg.board_step(n, d["boards"], d["turn"], d["remain"], d["pos"])
    out.update(b=b, t=t, r=r, w=w, e=e, mask=pkg.legal_mask(n, d["boards"]))
else:
    pol, val = pkg.net_forward(int(d["n"]), int(d["C"]), int(d["blocks"]), d["blob"], d["nb"], d["player"])
    out.update(pol=pol, val=val)
np.savez(sys.argv[2], **out)
""" % (ROOT, os.path.join(ROOT, "tests"))


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


def test_unfused_heads_match():
    need_gpu()
    import __graft_entry__ as ge
    n, C, blocks = 9, 64, 3
    rng = np.random.default_rng(5)
    blob = ge.make_random_blob(n, C, blocks, rng)
    nb = rng.integers(-1, 2, size=(37, n * n)).astype(np.int8)
    player = np.full(37, -1, np.int8)      # the oracle's forward takes one side to move per batch
    pol, val = pkg.net_forward(n, C, blocks, blob, nb, player)
    other = run_child({"C6_HEAD_FUSE": "0"}, n=n, C=C, blocks=blocks, blob=blob, nb=nb, player=player)
    # same products, different summation order of the 1x1 head convolution: tolerance of the net parity bar
    assert np.abs(other["pol"] - pol).max() < 1e-6 and np.abs(other["val"] - val).max() < 1e-5
    net = o.Net(n, C, blocks, blob)
    rpol, rval = net.forward(-1.0, nb.astype(np.float32))
    assert np.abs(pol - rpol).max() < 1e-5 and np.abs(val - rval).max() < 1e-5


def test_winograd_separate_transform_kernels_match():
    """C6_WINO_MID=0 C6_HEAD_FUSE=0 -- the FP32 Winograd F(4x4) tower with one input-transform and one output-transform kernel per
    layer and a separate head convolution, instead of the fused boundary kernel and the heads' filters in the last output transform
    (conv_wino4.hip): the same arithmetic in another order of kernels (model.hpp:61-92)."""
    need_gpu()
    import __graft_entry__ as ge
    n, C, blocks = 19, 256, 3
    rng = np.random.default_rng(6)
    blob = ge.make_random_blob(n, C, blocks, rng)
    nb = rng.integers(-1, 2, size=(70, n * n)).astype(np.int8)
    player = np.full(70, 1, np.int8)
    pol, val = pkg.net_forward(n, C, blocks, blob, nb, player)
    other = run_child({"C6_WINO_MID": "0", "C6_HEAD_FUSE": "0"}, n=n, C=C, blocks=blocks, blob=blob, nb=nb, player=player)
    assert np.abs(other["pol"] - pol).max() < 1e-6 and np.abs(other["val"] - val).max() < 1e-5
    other = run_child({"C6_WINO_STEM": "0"}, n=n, C=C, blocks=blocks, blob=blob, nb=nb, player=player)      # the stem as its own kernel
    assert np.abs(other["pol"] - pol).max() < 1e-6 and np.abs(other["val"] - val).max() < 1e-5
    net = o.Net(n, C, blocks, blob)
    for i in (0, 1, 35, 69):
        rpol, rval = net.forward(1.0, nb[i].astype(np.float32))
        assert np.abs(pol[i] - rpol).max() < 1e-5 and abs(val[i] - rval) < 1e-5


def test_winograd_weight_gradient_matches_direct_kernel():
    """FP32 training step at 256 channels: the weight gradient taken in the Winograd domain (dU = sum over tiles of A dY A^T x B^T x B,
    dg = G^T dU G; c6_conv_wino4_wgrad) against the direct kernel (C6_WGRAD_WINO=0) on the same batch -- 15x15 (mixed tiling) and 19x19,
    300 boards: every tile region holds whole and partial blocks and the reduction is cut into chunks.  Same sums, different order and
    different intermediate values: the whole gradient within 1e-5 in L2 (the trainer's parity bar against PyTorch is 1e-4)."""
    need_gpu()
    import __graft_entry__ as ge
    for n in (15, 19):
        C, blocks, B = 256, 2, 300
        rng = np.random.default_rng(7 + n)
        blob = ge.make_random_blob(n, C, blocks, rng)
        batch = dict(w=rng.integers(-1, 2, B).astype(np.float32), p=rng.choice([-1.0, 1.0], B).astype(np.float32),
                     b=rng.integers(-1, 2, (B, n * n)).astype(np.float32), q=rng.integers(0, n * n, B).astype(np.float32))
        a = run_child({"C6_WGRAD_WINO": "1"}, train=1, n=n, C=C, blocks=blocks, B=B, blob=blob, **batch)
        b = run_child({"C6_WGRAD_WINO": "0"}, train=1, n=n, C=C, blocks=blocks, B=B, blob=blob, **batch)
        assert abs(float(a["loss"]) - float(b["loss"])) <= 1e-6 * abs(float(b["loss"]))
        assert np.linalg.norm(a["grads"] - b["grads"]) <= 1e-5 * np.linalg.norm(b["grads"]), (n, np.linalg.norm(a["grads"] - b["grads"]) / np.linalg.norm(b["grads"]))
