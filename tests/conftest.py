import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import cpu_libs
    return cpu_libs.oracle()


@pytest.fixture(scope="session")
def reference():
    import cpu_libs
    ref = cpu_libs.reference()
    if ref is None:
        pytest.skip("reference library oracle/_ref/libref4x16.so not available")
    return ref
