"""K4 resident net (FP32 MFMA conv + heads) against the oracle's FP32 net and the committed PyTorch
golden vectors: |diff| <= 1e-5 on policy probabilities and value (BASELINE north_star tolerance)."""
import os

import numpy as np
import pytest

import c6oracle as o
from gpu_common import need_gpu, pkg

pytestmark = pytest.mark.gpu
TOL = 1e-5
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rand_blob(n, C, blocks, seed):
    import __graft_entry__ as ge
    return ge.make_random_blob(n, C, blocks, np.random.default_rng(seed))


# every parity-grade arithmetic of the 3x3 convolutions (C6_PREC_*): FP32 Winograd F(2x2,3x3) (the default), FP32 direct,
# split bf16, fp16 + MX-fp8
PRECISIONS = [pytest.param(1, id="fp32-winograd"), pytest.param(6, id="fp32-direct"), pytest.param(3, id="bf16x3-split"),
              pytest.param(5, id="f16f8-mixed")]


@pytest.mark.parametrize("prec", PRECISIONS)
@pytest.mark.parametrize("n,C,blocks,num", [(9, 64, 2, 9), (15, 64, 3, 5), (19, 128, 2, 4), (19, 256, 2, 3), (9, 256, 4, 8), (19, 256, 4, 2),
                                            (19, 256, 8, 2),      # BASELINE config 3's tower (stem + 7 blocks)
                                            # more boards than CUs, all distinct: persistent workgroups hand over from board to board
                                            # (double-buffered LDS slices, trickle across boards; Winograd blocks straddle boards)
                                            (19, 256, 2, 300), (15, 128, 3, 300), (9, 64, 2, 700),
                                            # Winograd F(4x4) (C = 256) at the other board sizes: 4x4 / 3x3 tiles per board, partial edge tiles,
                                            # 64-tile blocks that end inside a board and a last block that is not full
                                            (15, 256, 2, 7), (15, 256, 3, 150), (9, 256, 2, 100),
                                            # mixed tiling (19 = 4+4+4+4+3, 15 = 4+4+4+3): F(3,3) edge tiles live in regions of their own, each
                                            # with a last block of its own -- one board (every region a partial block), 13 boards (208 / 52 / 65
                                            # tiles: region B under one block), 65 boards of 15x15 (585 / 195 / 260)
                                            (19, 256, 2, 1), (19, 256, 3, 13), (15, 256, 2, 65)])
def test_net_forward_vs_oracle(n, C, blocks, num, prec):
    need_gpu()
    rng = np.random.default_rng(n * 1000 + C)
    blob = rand_blob(n, C, blocks, seed=n + C + blocks)
    boards = rng.integers(-1, 2, size=(num, n * n)).astype(np.int8)
    boards[0] = 0
    player = rng.choice(np.array([-1, 1], np.int8), size=num)
    pol, val = pkg.net_forward(n, C, blocks, blob, boards, player, precision=prec)
    net = o.Net(n, C, blocks, blob)
    check = range(num) if num <= 16 else sorted(i for i in set(list(range(6)) + [63, 64, 99, 100, 255, 256, 257, num // 2, num - 2, num - 1]) if i < num)
    for i in check:
        rp, rv = net.forward(float(player[i]), boards[i].astype(np.float32))
        assert np.abs(pol[i] - rp[0]).max() <= TOL, (i, np.abs(pol[i] - rp[0]).max())
        assert abs(val[i] - rv[0]) <= TOL
    assert np.allclose(pol.sum(axis=1), 1.0, atol=1e-5)
    net.close()


@pytest.mark.parametrize("prec", PRECISIONS)
def test_net_forward_large_activations(prec):
    """Activations in the hundreds (stem gain x300; the 1x1 head filters scaled back so that the outputs stay in their sensitive
    range).  The fp8 planes of the mixed path hold x_h / 8 (range 3584; a plain fp8 conversion turns values above 448 into NaN)
    and every layer's weights carry their own shift: same 1e-5 as at ordinary magnitudes."""
    need_gpu()
    n, C, blocks, num, gain = 19, 128, 3, 3, 300.0
    cells = n * n
    blob = rand_blob(n, C, blocks, seed=77).copy()
    stem_gamma = C * 18 + C
    blob[stem_gamma:stem_gamma + C] *= gain
    tower = C * 18 + C + 4 * C + 2 * (blocks - 1) * (C * C * 9 + C + 4 * C)
    blob[tower:tower + 2 * C] /= gain                                              # policy conv1x1 weights
    vconv = tower + 2 * C + 2 + 8 + cells * 2 * cells + cells
    blob[vconv:vconv + C] /= gain                                                  # value conv1x1 weights
    rng = np.random.default_rng(3)
    boards = rng.integers(-1, 2, size=(num, n * n)).astype(np.int8)
    pol, val = pkg.net_forward(n, C, blocks, blob, boards, 1, precision=prec)
    net = o.Net(n, C, blocks, blob)
    rp, rv = net.forward(1.0, boards.astype(np.float32))
    net.close()
    assert np.isfinite(pol).all() and np.isfinite(val).all()
    assert 0.05 < np.abs(rv).max() < 0.999 and rp.max() < 0.9                      # not saturated: the comparison means something
    assert np.abs(pol - rp).max() <= TOL and np.abs(val - rv).max() <= TOL, (np.abs(pol - rp).max(), np.abs(val - rv).max())


@pytest.mark.parametrize("n,C,blocks,num", [(19, 256, 4, 20), (15, 128, 3, 300), (9, 64, 3, 9)])
def test_relaxed_bf16_precision(n, C, blocks, num):
    """C6_PREC_BF16 (plain bf16 products) is NOT parity grade: its documented tolerance against the FP32
    oracle is 2e-3 on the value and 1e-4 on a policy probability (observed 1-3e-4 and 2-6e-6)."""
    need_gpu()
    rng = np.random.default_rng(n + 4 * C)
    blob = rand_blob(n, C, blocks, seed=n + C + blocks + 3)
    boards = rng.integers(-1, 2, size=(num, n * n)).astype(np.int8)
    player = rng.choice(np.array([-1, 1], np.int8), size=num)
    pol, val = pkg.net_forward(n, C, blocks, blob, boards, player, precision=pkg.PREC_BF16)
    net = o.Net(n, C, blocks, blob)
    worst = 0.0
    for i in list(range(min(num, 6))) + ([num - 1, 257] if num > 257 else []):
        rp, rv = net.forward(float(player[i]), boards[i].astype(np.float32))
        assert np.abs(pol[i] - rp[0]).max() <= 1e-4 and abs(val[i] - rv[0]) <= 2e-3
        worst = max(worst, abs(val[i] - rv[0]))
    assert worst > 1e-6          # it really is the relaxed path (the parity-grade paths sit below 1e-6 here)
    net.close()


@pytest.mark.parametrize("prec", PRECISIONS)
def test_net_forward_vs_torch_golden(prec):
    need_gpu()
    z = np.load(os.path.join(GOLD, "net_golden_n9_c64_b3.npz"))
    pol, val = pkg.net_forward(int(z["n"]), int(z["C"]), int(z["blocks"]), z["blob"], z["boards"], z["player"], precision=prec)
    assert np.abs(pol - z["policy"]).max() <= TOL and np.abs(val - z["value"]).max() <= TOL


@pytest.mark.parametrize("prec", PRECISIONS)
def test_net_forward_vs_torch_fp64_at_baseline_shape(prec):
    """HIP <-> PyTorch DIRECTLY at BASELINE config 2's net (19x19, C = 256, stem + 3 residual blocks): tests/torch_ref.py's
    restatement of model.hpp:9-95 on PyTorch's own CPU kernels in float64, built live (no 15 MB fixture), seeded weights with
    non-trivial BatchNorm statistics, against every parity-grade arithmetic at 1e-5."""
    need_gpu()
    import torch
    import torch_ref as tr
    n, C, blocks, num = 19, 256, 4, 2
    torch.manual_seed(1234)
    m = tr.WeightedPolicy(n, C, blocks)
    tr.randomize_bn(m)
    blob = tr.blob_of(m)
    rng = np.random.default_rng(11)
    boards = rng.integers(-1, 2, size=(num, n * n)).astype(np.int8)
    player = np.array([1, -1], np.int8)
    m = m.double().eval()
    with torch.no_grad():
        rp, rv = m(torch.from_numpy(player.astype(np.float64)), torch.from_numpy(boards.astype(np.float64)))
    rp, rv = rp.numpy(), rv.numpy().reshape(-1)
    pol, val = pkg.net_forward(n, C, blocks, blob, boards, player, precision=prec)
    assert np.abs(pol - rp).max() <= TOL and np.abs(val - rv).max() <= TOL, (np.abs(pol - rp).max(), np.abs(val - rv).max())


def test_all_paths_agree_at_benchmark_shape():
    """19x19, C = 256: Winograd FP32, the bf16x3 split and the mixed path agree with the direct FP32 MFMA path to the 1e-5 bar
    (the split error is relative, <= 3 * 2^-16 per product, and shrinks by 1/sqrt(K) in the 2304-term sums; Winograd's is a
    few f32 roundings in the transforms)"""
    need_gpu()
    n, C, blocks = 19, 256, 3
    blob = rand_blob(n, C, blocks, seed=77)
    rng = np.random.default_rng(5)
    boards = rng.integers(-1, 2, size=(3, n * n)).astype(np.int8)
    ref = pkg.net_forward(n, C, blocks, blob, boards, 1, precision=pkg.PREC_FP32_DIRECT)
    for prec in (pkg.PREC_FP32, pkg.PREC_BF16X3, pkg.PREC_F16F8):
        got = pkg.net_forward(n, C, blocks, blob, boards, 1, precision=prec)
        assert np.abs(ref[0] - got[0]).max() <= TOL and np.abs(ref[1] - got[1]).max() <= TOL, prec
    assert (pkg.net_forward(n, C, blocks, blob, boards, 1, precision=pkg.PREC_DEFAULT)[0] == pkg.net_forward(n, C, blocks, blob, boards, 1, precision=pkg.PREC_FP32)[0]).all()


@pytest.mark.parametrize("prec", PRECISIONS)
def test_engine_net_evaluator_matches_oracle_priors(prec):
    """the evaluator path inside the engine (dihedral gather -> net -> recover -> mask) equals get_from
    (mod.rs:218-234) computed by the oracle with its own FP32 net, to 1e-5"""
    need_gpu()
    n, C, blocks = 9, 64, 2
    blob = rand_blob(n, C, blocks, seed=4)
    eng = pkg.Engine(board_size=n, num_games=5, num_simulation=50, seed=1, evaluator=pkg.EVAL_NET, net_channels=C, net_blocks=blocks,
                     net_precision=prec)
    eng.load_weights(blob)
    eng.step(1)
    net = o.Net(n, C, blocks, blob)
    fn, ctx = net.eval_fn()
    az = o.AlphaZero(n, eval_kind=o.EVAL_CUSTOM, eval_fn=fn, ctx=ctx, seed=1, game_uid=0)
    s = o.sim_from_game(o.new_game(n))
    az.init(s)
    az.search(s)
    want = az.node_prob(np.zeros((n, n), np.int8)).reshape(-1)
    for g in range(5):
        r = eng.root(g)
        assert np.abs(r["prior"] - want).max() <= TOL
        assert abs(r["root_value"] - az.node_info(np.zeros((n, n), np.int8)).value) <= TOL
    eng.step(30)
    st = eng.stats()
    assert st["simulations"] == 31 * 5 and st["overflow_games"] == 0
    eng.close()
    net.close()


# ---- trained-like weights and the range guard of C6_PREC_F16F8 ---------------------------------------------------------

def _blob_offsets(n, C, blocks):
    """start of every BatchNorm gamma in the blob: [(name, offset, count)]"""
    cells = n * n
    out, off = [], C * 18 + C
    out.append(("stem", off, C)); off += 4 * C
    for l in range(2 * (blocks - 1)):
        off += C * C * 9 + C
        out.append((f"tower{l}", off, C)); off += 4 * C
    return out


@pytest.mark.parametrize("prec", PRECISIONS)
def test_net_after_training_steps(prec):
    """weights after 60 SGD steps of the HIP trainer (train mode moves every BatchNorm statistic away from its initial value,
    momentum spreads the channel scales): every parity-grade arithmetic still meets 1e-5 against the oracle's FP32 net"""
    need_gpu()
    import torch_ref as tr
    import torch
    n, C, blocks, B = 9, 64, 3, 64
    blob = rand_blob(n, C, blocks, seed=21)
    t = pkg.Trainer(n, C, blocks, B, lr=2e-2, momentum=0.9, precision=pkg.PREC_FP32)
    t.load_weights(blob)
    for k in range(60):
        w, p, b, q = [x.numpy() for x in tr.make_batch(n, B, 100 + k)]
        t.step(w, p, b, q)
    trained = t.export_weights()
    t.close()
    assert np.isfinite(trained).all() and np.abs(trained - blob).max() > 1e-3
    rng = np.random.default_rng(2)
    boards = rng.integers(-1, 2, size=(6, n * n)).astype(np.int8)
    player = rng.choice(np.array([-1, 1], np.int8), size=6)
    try:
        pol, val = pkg.net_forward(n, C, blocks, trained, boards, player, precision=prec)
    except pkg.C6Error as e:      # the bounded-range arithmetic may refuse a net -- loudly -- but never answers wrongly
        assert prec == pkg.PREC_F16F8 and f"error {pkg.ERR_RANGE}" in str(e)
        return
    net = o.Net(n, C, blocks, trained)
    for i in range(6):
        rp, rv = net.forward(float(player[i]), boards[i].astype(np.float32))
        assert np.abs(pol[i] - rp[0]).max() <= TOL and abs(val[i] - rv[0]) <= TOL
    net.close()


def _scaled_blob(n, C, blocks, spread, seed):
    """per-channel BatchNorm scales spanning `spread` decades around 1 in every tower layer, compensated in the NEXT layer's
    input weights so that the function stays well conditioned (what an optimiser does to a scale-invariant net): folded
    weights and activations then span the same decades"""
    blob = rand_blob(n, C, blocks, seed=seed).copy()
    rng = np.random.default_rng(seed)
    offs = _blob_offsets(n, C, blocks)
    L = 2 * (blocks - 1)
    # residual sums tie the scales of a block's output to its input: scale only the FIRST conv of each block (its output feeds
    # nothing but the block's second conv)
    for l in range(0, L, 2):
        name, goff, k = offs[1 + l]
        sc = 10.0 ** rng.uniform(-spread / 2, spread / 2, C).astype(np.float32)
        blob[goff:goff + C] *= sc                       # gamma
        blob[goff + C:goff + 2 * C] *= sc               # beta
        w2 = offs[1 + l][1] + 4 * C                     # next layer's conv weight [C_out, C_in, 3, 3]
        wv = blob[w2:w2 + C * C * 9].reshape(C, C, 9)
        wv /= sc[None, :, None]
    return blob


@pytest.mark.parametrize("prec", [pytest.param(1, id="fp32-winograd"), pytest.param(6, id="fp32-direct"), pytest.param(3, id="bf16x3-split")])
def test_net_channel_scales_spanning_six_decades(prec):
    """BatchNorm scales from 1e-3 to 1e3 inside every block: the full-range arithmetics (FP32 both ways, split bf16) hold 1e-5"""
    need_gpu()
    n, C, blocks = 15, 64, 3
    blob = _scaled_blob(n, C, blocks, spread=6.0, seed=5)
    rng = np.random.default_rng(8)
    boards = rng.integers(-1, 2, size=(4, n * n)).astype(np.int8)
    pol, val = pkg.net_forward(n, C, blocks, blob, boards, 1, precision=prec)
    net = o.Net(n, C, blocks, blob)
    rp, rv = net.forward(1.0, boards.astype(np.float32))
    net.close()
    assert np.abs(pol - rp).max() <= TOL and np.abs(val - rv).max() <= TOL, (np.abs(pol - rp).max(), np.abs(val - rv).max())


def test_f16f8_guard_rejects_weights_outside_its_range():
    """the same six-decade net on C6_PREC_F16F8: refused at load time with C6_ERR_RANGE (never wrong priors with C6_OK); a
    two-decade spread is inside the validated range and meets 1e-5"""
    need_gpu()
    n, C, blocks = 15, 64, 3
    boards = np.random.default_rng(8).integers(-1, 2, size=(4, n * n)).astype(np.int8)
    with pytest.raises(pkg.C6Error) as ei:
        pkg.net_forward(n, C, blocks, _scaled_blob(n, C, blocks, 6.0, 5), boards, 1, precision=pkg.PREC_F16F8)
    assert f"error {pkg.ERR_RANGE}" in str(ei.value) and "F16F8" in str(ei.value)
    eng = pkg.Engine(board_size=n, num_games=2, num_simulation=4, evaluator=pkg.EVAL_NET, net_channels=C, net_blocks=blocks, net_precision=pkg.PREC_F16F8)
    with pytest.raises(pkg.C6Error):
        eng.load_weights(_scaled_blob(n, C, blocks, 6.0, 5))
    with pytest.raises(pkg.C6Error):      # ... and the engine is left without weights, not with half-loaded ones
        eng.step(1)
    eng.close()
    blob = _scaled_blob(n, C, blocks, 2.0, 6)
    pol, val = pkg.net_forward(n, C, blocks, blob, boards, 1, precision=pkg.PREC_F16F8)
    net = o.Net(n, C, blocks, blob)
    rp, rv = net.forward(1.0, boards.astype(np.float32))
    net.close()
    assert np.abs(pol - rp).max() <= TOL and np.abs(val - rv).max() <= TOL


@pytest.mark.parametrize("gain,what", [(1e4, "3584"), (1e-4, "2^-3")])
def test_f16f8_guard_flags_activations_outside_its_range(gain, what):
    """stem activations far above 3584 (gain 1e4) or in fp16's subnormal neighbourhood (gain 1e-4): C6_PREC_F16F8 answers
    C6_ERR_RANGE (here already from the load-time bound of the stem; test_f16f8_engine_reports_range_error_while_playing covers
    the device-side flag); the full-range paths evaluate the same net to 1e-5"""
    need_gpu()
    n, C, blocks, num = 9, 64, 3, 3
    cells = n * n
    blob = rand_blob(n, C, blocks, seed=31).copy()
    # scale the stem's OUTPUT (gamma, beta) and compensate in the 1x1 head filters, as test_net_forward_large_activations does
    g0 = C * 18 + C
    blob[g0:g0 + 2 * C] *= gain
    tower = C * 18 + C + 4 * C + 2 * (blocks - 1) * (C * C * 9 + C + 4 * C)
    blob[tower:tower + 2 * C] /= gain
    vconv = tower + 2 * C + 2 + 8 + cells * 2 * cells + cells
    blob[vconv:vconv + C] /= gain
    boards = np.random.default_rng(3).integers(-1, 2, size=(num, cells)).astype(np.int8)
    with pytest.raises(pkg.C6Error) as ei:
        pkg.net_forward(n, C, blocks, blob, boards, 1, precision=pkg.PREC_F16F8)
    assert f"error {pkg.ERR_RANGE}" in str(ei.value) and what in str(ei.value)
    net = o.Net(n, C, blocks, blob)
    rp, rv = net.forward(1.0, boards.astype(np.float32))
    net.close()
    for prec in (pkg.PREC_FP32, pkg.PREC_FP32_DIRECT, pkg.PREC_BF16X3):
        pol, val = pkg.net_forward(n, C, blocks, blob, boards, 1, precision=prec)
        assert np.abs(pol - rp).max() <= TOL and np.abs(val - rv).max() <= TOL, (prec, np.abs(pol - rp).max(), np.abs(val - rv).max())


def test_f16f8_engine_reports_range_error_while_playing():
    """inside the engine the flag is sticky: step() itself never synchronises, the next synchronising call (stats) returns
    C6_ERR_RANGE instead of statistics built on wrong priors"""
    need_gpu()
    n, C, blocks = 9, 64, 3
    cells = n * n
    blob = rand_blob(n, C, blocks, seed=31).copy()
    # the residual stream grows by x30 per block: the stem passes the load-time bound, the second block's output does not
    offs = _blob_offsets(n, C, blocks)
    for l in (1, 3):
        goff = offs[1 + l][1]
        blob[goff:goff + 2 * C] *= 300.0
    eng = pkg.Engine(board_size=n, num_games=4, num_simulation=8, evaluator=pkg.EVAL_NET, net_channels=C, net_blocks=blocks,
                     net_precision=pkg.PREC_F16F8)
    eng.load_weights(blob)
    eng.step(3)
    with pytest.raises(pkg.C6Error) as ei:
        eng.stats()
    assert f"error {pkg.ERR_RANGE}" in str(ei.value)
    eng.close()


def test_blob_header_rejects_a_permuted_blob():
    """a blob with the right float count but libtorch's registration order (heads before the tower) used to load and
    play garbage.  blob_from_state_dict assembles by NAME; the header's checksum / geometry reject everything else."""
    need_gpu()
    import torch
    import torch_ref as tr
    n, C, blocks = 9, 64, 3
    torch.manual_seed(3)
    m = tr.WeightedPolicy(n, C, blocks)
    tr.randomize_bn(m)
    want = tr.blob_of(m)
    sd = {k: v for k, v in m.state_dict().items()}
    headed = pkg.blob_from_state_dict(sd, n, C, blocks)                  # ModuleList names (tower.i.j.*) via the alias table
    assert (headed[8:] == want).all()
    # the reference's own names (model.hpp:44-49: conv{i}_{j}, norm{i}_{j}), in libtorch's registration order: heads first
    ref_sd = {}
    for k, v in reversed(list(sd.items())):
        parts = k.split(".")
        if parts[0] == "tower":
            i, j = int(parts[1]), int(parts[2])
            k = ("conv" if j % 2 == 0 else "norm") + f"{i}_{j // 2}." + parts[3]
        ref_sd[k] = v
    assert (pkg.blob_from_state_dict(ref_sd, n, C, blocks)[8:] == want).all()
    boards = np.random.default_rng(1).integers(-1, 2, size=(2, n * n)).astype(np.int8)
    p1, v1 = pkg.net_forward(n, C, blocks, headed, boards, 1)
    p0, v0 = pkg.net_forward(n, C, blocks, want, boards, 1)
    assert (p1 == p0).all() and (v1 == v0).all()
    # a permuted payload under a valid header
    bad = headed.copy()
    k = C * 18
    bad[8:8 + k], bad[8 + k:8 + 2 * k] = headed[8 + k:8 + 2 * k].copy(), headed[8:8 + k].copy()
    with pytest.raises(pkg.C6Error) as ei:
        pkg.net_forward(n, C, blocks, bad, boards, 1)
    assert "checksum" in str(ei.value)
    with pytest.raises(pkg.C6Error) as ei:
        pkg.net_forward(n, C, blocks + 1, np.concatenate([headed, np.zeros(pkg.blob_floats(n, C, blocks + 1) - want.size, np.float32)]), boards, 1)
    assert "another board size" in str(ei.value) or "payload" in str(ei.value)


# ---- the BENCHMARK's own convolution kernels on hostile weights ---------------------------------------
# conv_wino6.hip (F(6,3) across) runs only at N = 19, C = 256, >= 3 blocks; conv_wino4.hip's F(4x4) at C = 256 for 15x15 / 9x9.  The
# robustness cases above sit on the C = 64 / 128 F(2x2) kernel, so every one of them is repeated here on the shapes the bench
# (config 2: (19, 256, 4); config 3: (19, 256, 8)) and the other C = 256 towers run.

BENCH_SHAPES = [pytest.param(19, 256, 4, id="w6-config2-net"), pytest.param(19, 256, 8, id="w6-config3-net"),
                pytest.param(15, 256, 3, id="w4-15x15"), pytest.param(9, 256, 4, id="w4-9x9")]


def _check_against_oracle(n, C, blocks, blob, boards, player, idx=None, prec=None):
    pol, val = pkg.net_forward(n, C, blocks, blob, boards, player, precision=pkg.PREC_FP32 if prec is None else prec)
    assert np.isfinite(pol).all() and np.isfinite(val).all()
    net = o.Net(n, C, blocks, blob)
    worst_p = worst_v = 0.0
    pl = np.broadcast_to(np.asarray(player, np.int8), (boards.shape[0],))
    for i in (range(boards.shape[0]) if idx is None else idx):
        rp, rv = net.forward(float(pl[i]), boards[i].astype(np.float32))
        worst_p = max(worst_p, float(np.abs(pol[i] - rp[0]).max()))
        worst_v = max(worst_v, float(abs(val[i] - rv[0])))
    net.close()
    print(f"[net-vs-oracle] n={n} C={C} blocks={blocks} boards={boards.shape[0]}: worst |policy diff| {worst_p:.3e}, worst |value diff| {worst_v:.3e}")      # (pytest -s: profiles/r04_hostile_errors.txt)
    assert worst_p <= TOL and worst_v <= TOL, (worst_p, worst_v)
    return pol, val, worst_p, worst_v


@pytest.mark.parametrize("n,C,blocks", BENCH_SHAPES)
def test_bench_kernels_channel_scales_spanning_six_decades(n, C, blocks):
    """(a) BatchNorm scales from 1e-3 to 1e3 inside every block, on F(6,3) / F(4x4) at C = 256"""
    need_gpu()
    blob = _scaled_blob(n, C, blocks, spread=6.0, seed=50 + n + blocks)
    boards = np.random.default_rng(80 + n).integers(-1, 2, size=(4, n * n)).astype(np.int8)
    player = np.array([1, -1, 1, -1], np.int8)
    _check_against_oracle(n, C, blocks, blob, boards, player)


@pytest.mark.parametrize("n,C,blocks", BENCH_SHAPES)
def test_bench_kernels_large_activations(n, C, blocks):
    """(b) activations in the hundreds (stem gain x300, compensated in the heads' 1x1 filters) on F(6,3) / F(4x4) at C = 256"""
    need_gpu()
    gain, cells = 300.0, n * n
    blob = rand_blob(n, C, blocks, seed=77 + n + blocks).copy()
    stem_gamma = C * 18 + C
    blob[stem_gamma:stem_gamma + C] *= gain
    tower = C * 18 + C + 4 * C + 2 * (blocks - 1) * (C * C * 9 + C + 4 * C)
    blob[tower:tower + 2 * C] /= gain
    vconv = tower + 2 * C + 2 + 8 + cells * 2 * cells + cells
    blob[vconv:vconv + C] /= gain
    boards = np.random.default_rng(3 + n).integers(-1, 2, size=(3, cells)).astype(np.int8)
    _check_against_oracle(n, C, blocks, blob, boards, np.array([1, -1, 1], np.int8))


@pytest.mark.parametrize("n,C,blocks", BENCH_SHAPES)
def test_bench_kernels_after_training_steps(n, C, blocks):
    """(c) weights after 60 SGD steps of the HIP trainer AT C = 256 (every BatchNorm statistic moved, momentum spread the channel
    scales), evaluated by the benchmark's kernels against the oracle's FP32 net"""
    need_gpu()
    import torch_ref as tr
    B = 32
    blob = rand_blob(n, C, blocks, seed=21 + n + blocks)
    t = pkg.Trainer(n, C, blocks, B, lr=1e-2, momentum=0.9, precision=pkg.PREC_FP32)
    t.load_weights(blob)
    for k in range(60):
        w, p, b, q = [x.numpy() for x in tr.make_batch(n, B, 300 + k)]
        t.step(w, p, b, q)
    trained = t.export_weights()
    t.close()
    assert np.isfinite(trained).all() and np.abs(trained - blob).max() > 1e-3
    rng = np.random.default_rng(2 + n)
    boards = rng.integers(-1, 2, size=(5, n * n)).astype(np.int8)
    boards[0] = 0
    player = rng.choice(np.array([-1, 1], np.int8), size=5)
    _check_against_oracle(n, C, blocks, trained, boards, player)


@pytest.mark.parametrize("n,C,blocks,num", [pytest.param(19, 256, 4, 333, id="w6-333-boards-every-region-partial"),
                                            pytest.param(19, 256, 8, 320, id="w6-320-boards-every-region-exactly-full"),
                                            pytest.param(15, 256, 3, 301, id="w4-15x15-301"), pytest.param(9, 256, 4, 500, id="w4-9x9-500")])
def test_bench_kernels_many_distinct_boards(n, C, blocks, num):
    """(d) >= 300 DISTINCT boards in one net_forward: every tile class (w6: D 10, A 4, B 4, B' 2 tiles per board in regions of
    their own, 64-tile blocks) has full blocks and -- at 333 -- a partial last block, blocks straddle boards; >= 16 sampled
    boards (first / last, the block seams of every class) against the oracle; boards of every density from empty to full"""
    need_gpu()
    rng = np.random.default_rng(1000 + num)
    dens = rng.uniform(0, 1, size=(num, 1))
    u = rng.uniform(0, 1, size=(num, n * n))
    boards = np.where(u < dens, rng.choice(np.array([-1, 1], np.int8), size=(num, n * n)), 0).astype(np.int8)
    boards[0] = 0
    boards[1] = 1
    boards[2] = -1
    seen = set()
    for i in range(num):                                  # sparse boards repeat: redraw until all are distinct
        while boards[i].tobytes() in seen:
            boards[i] = rng.integers(-1, 2, size=n * n)
        seen.add(boards[i].tobytes())
    assert len(seen) == num
    player = rng.choice(np.array([-1, 1], np.int8), size=num)
    blob = rand_blob(n, C, blocks, seed=5 + num)
    # seams: 64-tile blocks end inside boards 6 / 12 / 16 / 32 (classes of 10 / 4 / 4 / 2 tiles per board at 19x19)
    idx = sorted({0, 1, 2, 5, 6, 7, 12, 13, 15, 16, 31, 32, 63, 64, 160, 255, 256, num // 2, num - 3, num - 2, num - 1})
    assert len(idx) >= 16
    pol, _, _, _ = _check_against_oracle(n, C, blocks, blob, boards, player, idx=idx)
    assert np.allclose(pol.sum(axis=1), 1.0, atol=1e-5)
    # every board's answer is independent of the batch it travels in: the sampled ones again, alone, bit for bit
    p1, v1 = pkg.net_forward(n, C, blocks, blob, boards[idx[-4:]], player[idx[-4:]], precision=pkg.PREC_FP32)
    assert (p1 == pol[idx[-4:]]).all()
