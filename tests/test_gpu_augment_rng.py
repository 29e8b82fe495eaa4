"""K3 augment/recover and the C6RNG Dirichlet noise on the GPU against the oracle: bit-exact.
Reference: alphazero_policy/augment.rs:60-92, mod.rs:218-234, mod.rs:260-262."""
import numpy as np
import pytest

import c6oracle as o
from gpu_common import need_gpu, pkg

pytestmark = pytest.mark.gpu


def test_augment8(n):
    need_gpu()
    rng = np.random.default_rng(n)
    boards = rng.integers(-1, 2, size=(33, n, n)).astype(np.int8)
    got = pkg.augment8(n, boards)
    for b, g in zip(boards, got):
        assert (o.augment_way8(b) == g).all()


def test_recover8_bit_exact(n):
    need_gpu()
    rng = np.random.default_rng(10 + n)
    num = 29
    boards = rng.integers(-1, 2, size=(num, n, n)).astype(np.int8)
    v8 = rng.uniform(-1, 1, size=(num, 8)).astype(np.float32)
    p8 = rng.uniform(-1, 1, size=(num, 8, n, n)).astype(np.float32)
    val, pri = pkg.recover8(n, boards, v8, p8)
    for i in range(num):
        want = o.recover_way8(p8[i]) * (boards[i] == 0)
        assert (pri[i] == want).all()
        s = np.float32(0)
        for k in range(8):
            s = np.float32(s + v8[i, k])
        assert val[i] == np.float32(s / np.float32(8))


def test_augment_recover_round_trip(n):  # augment_tests.rs:114-142 on the GPU
    need_gpu()
    rng = np.random.default_rng(20 + n)
    boards = rng.integers(-1, 2, size=(5, n, n)).astype(np.int8)
    aug = pkg.augment8(n, boards)
    val, pri = pkg.recover8(n, np.zeros_like(boards), np.zeros((5, 8), np.float32), aug.astype(np.float32))
    assert (pri == boards.astype(np.float32)).all()


def test_dirichlet_noise_bit_exact(n):
    need_gpu()
    rng = np.random.default_rng(30 + n)
    num = 24
    boards = (rng.uniform(size=(num, n, n)) < 0.3).astype(np.int8) * rng.choice(np.array([-1, 1], np.int8), size=(num, n, n))
    k0 = rng.integers(0, 2**32, num, dtype=np.uint64).astype(np.uint32)
    k1 = rng.integers(0, 2**32, num, dtype=np.uint64).astype(np.uint32)
    search = rng.integers(0, 100000, num).astype(np.uint32)
    depth = rng.integers(0, 6, num).astype(np.uint32)
    for alpha in (0.03, 0.5):
        got = pkg.dirichlet_noise(n, alpha, k0, k1, search, depth, boards)
        for i in range(num):
            want = o.dirichlet_noise(alpha, int(k0[i]), int(k1[i]), int(search[i]), int(depth[i]), boards[i])
            assert (got[i] == want).all(), (alpha, i, np.abs(got[i] - want).max())
