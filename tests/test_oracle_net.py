"""Pins the oracle's C net to the committed PyTorch golden vectors (tests/golden/gen_net_golden.py)."""
import os

import numpy as np

import c6oracle as o

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_oracle_net_vs_torch_golden():
    z = np.load(os.path.join(GOLD, "net_golden_n9_c64_b3.npz"))
    n, C, blocks = int(z["n"]), int(z["C"]), int(z["blocks"])
    net = o.Net(n, C, blocks, z["blob"])
    for i in range(z["boards"].shape[0]):
        pol, val = net.forward(float(z["player"][i]), z["boards"][i].astype(np.float32))
        assert np.abs(pol[0] - z["policy"][i]).max() <= 1e-5
        assert abs(val[0] - z["value"][i]) <= 1e-5
    net.close()


def test_blob_size_formula():
    # model.hpp:10-50: parameter/buffer count of the reference architecture
    n, C, nb = 19, 256, 4
    n2 = n * n
    want = (C * 18 + C + 4 * C) + (nb - 1) * 2 * (C * C * 9 + C + 4 * C) + (2 * C + 2 + 8) + (n2 * 2 * n2 + n2) \
        + (C + 1 + 4 * n2) + (256 * n2 + 256) + (256 + 1)
    assert o.lib().c6o_net_blob_floats(n, C, nb) == want


def test_oracle_net_vs_live_torch_float64_at_other_shapes():
    """r5: beside the golden vectors (one shape), the oracle's f32 net against the PyTorch restatement of model.hpp:9-95 in float64, built LIVE at the
    other board sizes -- 15x15 (the reference's compiled size) and 19x19 (the benchmark's), randomized BatchNorm statistics, boards of both sides to move --
    within the parity bar of the GPU tests (1e-5 on every policy entry and on the value)."""
    import torch

    import torch_ref as tr
    for n, C, blocks, seed in ((15, 32, 3, 5), (19, 16, 2, 6), (9, 32, 4, 7)):
        torch.manual_seed(seed)
        m = tr.WeightedPolicy(n, C, blocks)
        tr.randomize_bn(m)
        blob = tr.blob_of(m)
        m = m.double().eval()
        rng = np.random.default_rng(seed)
        boards = rng.integers(-1, 2, size=(6, n * n)).astype(np.float32)
        net = o.Net(n, C, blocks, blob)
        for player in (-1.0, 1.0):
            with torch.no_grad():
                pol_t, val_t = m(torch.full((boards.shape[0],), player, dtype=torch.float64), torch.from_numpy(boards).double())
            pol, val = net.forward(player, boards)
            assert np.abs(pol - pol_t.numpy()).max() <= 1e-5, (n, C, blocks, player)
            assert np.abs(val - val_t.numpy().reshape(-1)).max() <= 1e-5, (n, C, blocks, player)
            assert abs(float(pol.sum(1).mean()) - 1.0) < 1e-5
        net.close()
