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
