import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from c6amd_loader import load_package  # noqa: E402

pkg = load_package()


def need_gpu():
    if pkg.device_count() < 1:
        pytest.fail("no MI355X visible: -m gpu tests need the GPU box (libconnect6_amd has no CPU fallback)")
