import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for path in (ROOT, os.path.join(ROOT, "tests")):
    if path not in sys.path:
        sys.path.insert(0, path)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_ffi

    oracle_ffi.load()
    return oracle_ffi


@pytest.fixture(scope="session")
def ffi():
    from rt_amd import ffi as _ffi

    return _ffi


@pytest.fixture(scope="session")
def hip(ffi):
    """The HIP renderer.  No fallback: without librt_hip.so or a GPU this raises and the test fails."""
    renderer = ffi.HipRenderer(0)
    yield renderer
    renderer.close()


@pytest.fixture(scope="session")
def scenes(ffi):
    cache = {}

    def get(index, seed=1, spheres=10000):
        key = (index, seed, spheres)
        if key not in cache:
            cache[key] = ffi.Scene.demo(index, seed, spheres)
        return cache[key]

    return get
