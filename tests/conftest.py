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


class _Options:
    """Library options for the duration of a test (include/rans4x16_hip.h part 2b): set as the process-wide default -
    what the five drop-in symbols and contexts created from now on use - AND on the calling thread's context (the
    host-batch helpers of htscodecs_amd.codec); restored afterwards."""

    def __init__(self):
        self.saved = {}

    def set(self, name, value):
        from htscodecs_amd import codec
        if name not in self.saved:
            self.saved[name] = codec.get_default_option(name)
        codec.set_default_option(name, int(value))
        codec.set_option(name, int(value))

    def restore(self):
        from htscodecs_amd import codec
        for name, value in self.saved.items():
            codec.set_default_option(name, value)
            codec.set_option(name, value)
        self.saved = {}


@pytest.fixture
def opts():
    o = _Options()
    yield o
    o.restore()
