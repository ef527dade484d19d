import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Compile every native artefact once per session (hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as g
    g.build()
    return True


@pytest.fixture(scope="session")
def scenes_dir():
    return os.path.join(ROOT, "scenes")


def f2u(x):
    """uint32 values carried in a float32 array (bit pattern preserved)."""
    import numpy as np
    return np.asarray(x, np.uint32).view(np.float32)


def rel_l2(a, b):
    import numpy as np
    a = a[..., :3].astype(np.float64)
    b = b[..., :3].astype(np.float64)
    return float(np.sqrt(((a - b) ** 2).sum() / max((b ** 2).sum(), 1e-30)))
