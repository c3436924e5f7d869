import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_rgb():
    from oracle import binding
    return binding.load("oracle")


@pytest.fixture(scope="session")
def ref_rgb():
    """The compiled reference (oracle/_ref); tests needing it skip where it was not built."""
    from oracle import binding
    lib = binding.load("ref_rgb")
    if lib is None:
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    return lib


@pytest.fixture(scope="session")
def oracle_spectral():
    from oracle import binding
    from slr_amd import abi
    return binding.load("oracle", abi.MODE_SPECTRAL)


@pytest.fixture(scope="session")
def ref_spectral():
    from oracle import binding
    lib = binding.load("ref_spectral")
    if lib is None:
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    return lib
