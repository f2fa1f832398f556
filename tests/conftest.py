import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
REF_IMG = "/root/reference/img"  # only present in the build container, never on the GPU box


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure, oracle/stm_oracle.c)."""
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def stm():
    """The product package (host mirror of the C ABI).  If the HIP library has not been built in this tree yet it is
    BUILT (hipcc, gfx950) -- never substituted: there is no other implementation to fall back to."""
    import stm_amd
    if not os.path.exists(stm_amd.LIB_PATH):
        stm_amd.build()
    return stm_amd


@pytest.fixture(scope="session")
def golden():
    return dict(np.load(os.path.join(GOLDEN, "bud_crop_golden.npz")))


@pytest.fixture(scope="session")
def gpu_ready(stm):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("a -m gpu test ran without a GPU")
    stm.lib()  # raises loudly if libstm_hip.so is missing: no fallback
    return True


def rand_pair(H, W, seed, smooth=True):
    """Small random stereo-ish pair: smooth base + noise so arms have a realistic length mix."""
    rng = np.random.RandomState(seed)
    base = rng.randint(0, 256, size=(H // 8 + 2, W // 8 + 2, 3)).astype(np.float32)
    img = np.kron(base, np.ones((8, 8, 1), np.float32))[:H, :W]
    if smooth:
        img = img * 0.8 + rng.randint(0, 52, size=(H, W, 3))
    L = np.clip(img, 0, 255).astype(np.uint8)
    R = np.roll(L, -3, axis=1).copy()
    R[:, ::7] = np.clip(R[:, ::7].astype(np.int32) + rng.randint(-9, 10, size=R[:, ::7].shape), 0, 255).astype(np.uint8)
    return L, R
