import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from blasr_testlib import load_oracle
    return load_oracle()


@pytest.fixture(scope="session")
def ref():
    from blasr_testlib import load_ref
    r = load_ref()
    if r is None:
        pytest.skip("oracle/_ref/libblasr_ref.so not built (needs /root/reference)")
    return r


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "refine_golden.npz"))
