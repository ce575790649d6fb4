import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "go-jpeg2000_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


# the tests' A/B switches (J2K_L0_WG, J2K_T1_DEC_SPLIT, ...) are environment variables of the TUNING set: the library reads them only
# when J2K_TUNING=1 is there (a host process does not inherit kernel choices from its environment); set before the library loads
os.environ.setdefault("J2K_TUNING", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    import oracle as o
    o.lib()
    return o
