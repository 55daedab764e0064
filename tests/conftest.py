import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "slow: long-running CPU check")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def hip():
    """The HIP backend through its C-ABI.  Fails loudly (no skip) when a GPU test is
    selected but the library or the device is missing."""
    import pedoni_amd
    from pedoni_amd import abi
    abi.load_library()
    assert abi.device_count() >= 1, "gpu test selected but no HIP device is visible"
    return pedoni_amd
