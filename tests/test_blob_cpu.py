"""Weight-blob layout, converter and header (host only).  ADVICE r1: a blob in libtorch's registration order has the right float
count and used to load; the converter assembles BY NAME and the header's geometry + checksum reject the rest."""
import numpy as np
import pytest

from gpu_common import pkg


def _model(n=9, C=64, blocks=3, seed=3):
    import torch
    import torch_ref as tr
    torch.manual_seed(seed)
    m = tr.WeightedPolicy(n, C, blocks)
    tr.randomize_bn(m)
    return m, tr.blob_of(m)


def test_blob_from_state_dict_is_name_driven():
    n, C, blocks = 9, 64, 3
    m, want = _model(n, C, blocks)
    sd = dict(m.state_dict())
    headed = pkg.blob_from_state_dict(sd, n, C, blocks)
    assert headed.size == pkg.blob_floats(n, C, blocks) + 8 and (headed[8:] == want).all()
    assert (pkg.blob_from_state_dict(sd, n, C, blocks, header=False) == want).all()
    # the reference's own module names (model.hpp:44-49 conv{i}_{j} / norm{i}_{j}) in REVERSED order: same blob
    ref_sd = {}
    for k, v in reversed(list(sd.items())):
        parts = k.split(".")
        if parts[0] == "tower":
            i, j = int(parts[1]), int(parts[2])
            k = ("conv" if j % 2 == 0 else "norm") + f"{i}_{j // 2}." + parts[3]
        ref_sd[k] = v
    assert (pkg.blob_from_state_dict(ref_sd, n, C, blocks)[8:] == want).all()
    del ref_sd["value_norm.running_var"]
    with pytest.raises(pkg.C6Error, match="lacks value_norm.running_var"):
        pkg.blob_from_state_dict(ref_sd, n, C, blocks)
    sd["policy_linear.weight"] = sd["policy_linear.weight"].T
    with pytest.raises(pkg.C6Error, match="policy_linear.weight: shape"):
        pkg.blob_from_state_dict(sd, n, C, blocks)


def test_header_rejects_permutation_geometry_and_corruption(tmp_path):
    n, C, blocks = 9, 64, 3
    m, want = _model(n, C, blocks)
    headed = pkg.add_blob_header(want, n, C, blocks)
    assert (pkg.check_blob(headed, n, C, blocks) == want).all()
    k = C * 18
    bad = headed.copy()
    bad[8:8 + k], bad[8 + k:8 + 2 * k] = headed[8 + k:8 + 2 * k].copy(), headed[8:8 + k].copy()      # two tensors' worth swapped: same size
    with pytest.raises(pkg.C6Error, match="checksum"):
        pkg.check_blob(bad, n, C, blocks)
    flip = headed.copy()
    flip[1000] = np.nextafter(flip[1000], np.float32(1e9))
    with pytest.raises(pkg.C6Error, match="checksum"):
        pkg.check_blob(flip, n, C, blocks)
    with pytest.raises(pkg.C6Error, match="another board size"):
        pkg.check_blob(headed, n, C * 2, blocks)
    with pytest.raises(pkg.C6Error, match="magic"):
        pkg.check_blob(np.concatenate([np.zeros(8, np.float32), want]), n, C, blocks)
    for k, v in ((5, np.nan), (6, -1.0), (5, 3.0e38), (6, 0.5), (7, np.inf)):      # size / checksum fields that are not 24-bit integers
        junk = headed.copy()
        junk[k] = v
        with pytest.raises(pkg.C6Error, match="corrupted header"):
            pkg.check_blob(junk, n, C, blocks)
    path = str(tmp_path / "net.c6w")
    pkg.save_blob(path, dict(m.state_dict()), n, C, blocks)
    assert (pkg.load_blob(path, n, C, blocks)[8:] == want).all()
