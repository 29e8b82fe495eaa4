"""Error budget of the default convolution arithmetic (C6_PREC_F16F8, conv_m8.hip), emulated on the CPU: fp16 main term +
OCP-e4m3 cross terms with the shipped shifts, float64 accumulation, through a small policy/value net of the reference's
architecture (AlphaZero/cpp_weighted/model.hpp:52-95).  The GPU tests hold the kernel to 1e-5 against the oracle; this one
states how far inside that bar the arithmetic itself sits, beside the bf16 split's.  No GPU needed."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
import mixed_err


def test_mixed_product_error_budget():
    res = mixed_err.run(9, 64, 3, 3, modes=("bf16x3", "mixed"))
    assert 0.5 < res["_max_activation"] < 100.0
    for mode in ("bf16x3", "mixed"):
        pol, val, act = res[mode]
        assert pol < 1e-7 and val < 2e-6, (mode, res[mode])        # 1e-5 is the parity bar
    # same accuracy class: the mixed scheme's error stays within a small factor of the bf16 split's
    assert res["mixed"][1] < 20 * max(res["bf16x3"][1], 1e-8)
