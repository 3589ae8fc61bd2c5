import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture
def devlib():
    """the development flavour of the library (-DDFA_DEV_AB, dynfu_amd/libdynfu_amd_dev.so): the DFA_* environment A/B
    switches and the non-default kernel variants exist there only — tests that compare variants run inside it"""
    from dynfu_amd import _lib
    with _lib.use_library(_lib.dev_lib_path()):
        yield
