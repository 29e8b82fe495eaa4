"""Host-side weight image of the mixed-precision convolution (C6_PREC_F16F8, conv_m8.hip): fp16 / OCP-e4m3 rounding and the
operand layout of the two MFMAs, against NumPy fp16 and PyTorch's float8_e4m3fn cast.  No GPU needed.
(The arithmetic it serves: AlphaZero/cpp_weighted/model.hpp:64-74, the 3x3 convolutions of the tower.)"""
import numpy as np
import pytest
import torch

from gpu_common import pkg


def e4m3(x):
    return torch.tensor(np.asarray(x, np.float32)).to(torch.float8_e4m3fn).view(torch.uint8).numpy()


@pytest.mark.parametrize("C,big", [(64, 13.9), (128, 0.11), (64, 700.0)])
def test_pack_m8_layout_and_rounding(C, big):
    rng = np.random.default_rng(C + int(big))
    w = (rng.standard_normal((C, C, 3, 3)) * 10.0 ** rng.uniform(-6, -0.5, (C, C, 3, 3))).astype(np.float32)
    w[0, 0, 0, 0] = 0.0
    w[1, 2, 1, 1] = big        # sets the layer's weight shift d
    img = pkg.conv_pack_m8(C, w)
    KS = WAVES = C // 32
    n16 = KS * 10 * WAVES * 2 * 64 * 8
    d = int(img[:4].view(np.int32)[0])      # 16-byte header: the largest shift that keeps max |w_h| 2^d below 448 (416 with headroom)
    wt = w.reshape(C, C, 9)
    wh = wt.astype(np.float16)
    wl = wt - wh.astype(np.float32)
    wmax = float(np.abs(wh.astype(np.float32)).max())
    assert d == int(np.clip(np.floor(np.log2(416.0 / wmax)), -16, 24)) and (img[4:16] == 0).all()
    assert wmax * 2.0 ** d <= 416.0 < wmax * 2.0 ** (d + 1) or d in (-16, 24)
    p16 = img[16:16 + n16 * 2].view(np.float16).reshape(KS, 10, WAVES, 2, 64, 8)
    p8 = img[16 + n16 * 2:].reshape(KS, 5, WAVES, 2, 64, 32)
    assert img.size == 16 + n16 * 2 + p8.size
    lane = np.arange(64)
    for ks in range(KS):
        for wave in range(WAVES):
            for ct in range(2):
                n = wave * 32 + ct * 16 + (lane & 15)                        # output channel of the lane's row
                k16 = ks * 32 + (lane >> 4)[:, None] * 8 + np.arange(8)       # fp16 plane: 8 input channels of k-block lane >> 4
                for tap in range(9):
                    assert (p16[ks, tap, wave, ct] == wh[n[:, None], k16, tap]).all()
                assert (p16[ks, 9, wave, ct] == 0).all()                      # the zero tap beside the lone tap 8
                g = lane >> 4
                k8 = ks * 32 + 16 * (g & 1)[:, None] + np.arange(16)          # fp8 planes: 16 input channels of half g & 1
                for pair in range(5):
                    tap = 2 * pair + (g >> 1)                                 # groups 0-1: tap 2 pair, groups 2-3: tap 2 pair + 1
                    ok = tap < 9
                    t = np.minimum(tap, 8)
                    lo = np.where(ok[:, None], e4m3(np.ldexp(wl[n[:, None], k8, t[:, None]], d + 11)), 0)
                    hi = np.where(ok[:, None], e4m3(np.ldexp(wh[n[:, None], k8, t[:, None]].astype(np.float32), d)), 0)
                    got = p8[ks, pair, wave, ct]
                    assert (got[:, :16] == lo).all() and (got[:, 16:] == hi).all()


def test_pack_m8_rejects_bad_sizes():
    L = pkg.lib()
    assert L.c6_conv_pack_m8_bytes(100) == 0
    w = np.zeros((64, 64, 3, 3), np.float32)
    out = np.zeros(16, np.uint8)
    import ctypes as C
    assert L.c6_conv_pack_m8(64, w.ctypes.data_as(C.POINTER(C.c_float)), out.ctypes.data_as(C.POINTER(C.c_uint8)), 16) != 0
