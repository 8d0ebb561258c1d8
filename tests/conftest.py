import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import lldo
    lldo.lib()
    return lldo


@pytest.fixture(scope="session")
def golden_synth():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "mfcc12_0_d_a_synth.npz"))


@pytest.fixture(scope="session")
def golden_config1():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "mfcc12_0_d_a_config1.npz"))
