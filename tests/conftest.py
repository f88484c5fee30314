import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture
def lib_option():
    """Sets documented run-time options of the library (tf_set_option) and restores them afterwards."""
    from transflow_amd import _lib
    saved = {}

    def set_(name, value):
        saved.setdefault(name, _lib.get_option(name))
        _lib.set_option(name, value)
    yield set_
    for name, value in saved.items():
        _lib.set_option(name, value)
