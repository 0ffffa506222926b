import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "aloception-oss_amd"), os.path.join(ROOT, "oracle"), ROOT, os.path.dirname(__file__)):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name))

    return load


@pytest.fixture(scope="session", autouse=True)
def _built_oracle():
    import oracle  # noqa: F401  (oracle/oracle.py; builds libalo_oracle.so with gcc when it is missing)

    oracle.build()
