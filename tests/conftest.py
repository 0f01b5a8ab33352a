import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle64():
    from oracle.orc import Oracle, build

    build()
    return Oracle("f64")


@pytest.fixture(scope="session")
def oracle32():
    from oracle.orc import Oracle, build

    build()
    return Oracle("f32")


@pytest.fixture(scope="session")
def renderer():
    """A context on GPU 0 through the C ABI; fails loudly when the library or the GPU is missing."""
    from raytracer_amd import Renderer

    r = Renderer(0)
    yield r
    r.close()
