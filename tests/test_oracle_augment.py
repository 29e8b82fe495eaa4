"""K-7: Connect6/libconnect6/src/policy/alphazero_policy/augment_tests.rs:4-142 on the oracle."""
import ctypes as C

import numpy as np

import c6oracle as o

B, Z, W = o.BLACK, o.NONE, o.WHITE


def corner_board(n):
    b = np.zeros((n, n), np.int8)
    b[0, 0] = B
    b[0, 1] = B
    b[0, n - 1] = W
    b[1, n - 1] = B
    b[n - 1, n - 1] = W
    b[n - 1, n - 2] = W
    b[n - 1, 0] = B
    b[n - 2, 0] = W
    return b


def apply(name, b):
    b = b.copy()
    getattr(o.lib(), f"c6o_{name}_i8")(o.i8(b), b.shape[0])
    return b


def test_rotate_left(n):  # augment_tests.rs:4-26
    b = apply("rotate_left", corner_board(n))
    assert b[n - 1, 0] == B and b[n - 2, 0] == B and b[0, 0] == W and b[0, 1] == B
    assert b[0, n - 1] == W and b[1, n - 1] == W and b[n - 1, n - 1] == B and b[n - 1, n - 2] == W
    assert (b == np.rot90(corner_board(n), 1)).all()


def test_rotate_right(n):  # augment_tests.rs:28-50
    b = apply("rotate_right", corner_board(n))
    assert b[n - 1, 0] == W and b[n - 2, 0] == W and b[0, 0] == B and b[0, 1] == W
    assert b[0, n - 1] == B and b[1, n - 1] == B and b[n - 1, n - 1] == W and b[n - 1, n - 2] == B
    assert (b == np.rot90(corner_board(n), -1)).all()


def test_flip_vertical(n):  # augment_tests.rs:52-72
    b = apply("flip_vertical", corner_board(n))
    assert b[n - 1, 0] == W and b[n - 1, 1] == W and b[0, 0] == W and b[1, 0] == B
    assert b[0, n - 1] == B and b[0, n - 2] == B and b[n - 1, n - 1] == B and b[n - 2, n - 1] == W
    assert (b == np.fliplr(corner_board(n))).all()


def test_flip_horizontal(n):  # augment_tests.rs:74-94
    b = apply("flip_horizontal", corner_board(n))
    assert b[n - 1, 0] == B and b[n - 1, 1] == B and b[0, 0] == B and b[1, 0] == W
    assert b[0, n - 1] == W and b[0, n - 2] == W and b[n - 1, n - 1] == W and b[n - 2, n - 1] == B


def test_augment_and_recover(n):  # augment_tests.rs:114-142
    rng = np.random.default_rng(n)
    board = rng.integers(-1, 2, size=(n, n)).astype(np.int8)
    aug = o.augment_way8(board)
    for i in range(8):
        for j in range(i + 1, 8):
            assert (aug[i] != aug[j]).any()
    # image order [R1, F R1, R2, F R2, R3, F R3, id, F id] (augment.rs:60-73)
    for i in range(4):
        assert (aug[2 * i] == np.rot90(board, i + 1)).all()
        assert (aug[2 * i + 1] == np.fliplr(np.rot90(board, i + 1))).all()
    rec = o.recover_way8(aug.astype(np.float32))
    assert (rec == board.astype(np.float32)).all()


def test_recover_sum_order(n):
    # (((t0 + t1) + t2) + t3) / 8 with t_i = p[2i] + flipV(p[2i+1]) rotated back (augment.rs:79-91)
    rng = np.random.default_rng(100 + n)
    p = rng.uniform(-1, 1, size=(8, n, n)).astype(np.float32)
    total = np.zeros((n, n), np.float32)
    for i in range(4):
        t = (p[2 * i] + np.fliplr(p[2 * i + 1])).astype(np.float32)
        t = np.rot90(t, -(i + 1))
        total = (total + t).astype(np.float32)
    total = (total / np.float32(8)).astype(np.float32)
    assert (o.recover_way8(p) == total).all()
