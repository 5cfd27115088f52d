import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN

# let small test shapes reach the direct 3x3x3 kernel (the product default keeps it for grids that fill the chip)
import os as _os
_os.environ.setdefault("OTAL_CONV_DIRECT_MINTILES", "1")
