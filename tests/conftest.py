import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def zk():
    """The product package (loads libzkprove.so; fails loudly if the HIP library is missing)."""
    return importlib.import_module("zk-symmetric-crypto_amd")


@pytest.fixture(scope="session")
def oracle():
    from oracle import ref
    ref.lib()
    ref.set_threads(min(8, os.cpu_count() or 1))
    return ref
