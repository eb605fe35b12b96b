import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session")
def port_oracle():
    from oracle.oracle import Oracle
    return Oracle("port")


@pytest.fixture(scope="session")
def ref_oracle():
    from oracle import oracle
    if not oracle.have("reference") and not os.path.exists("/root/reference"):
        pytest.skip("oracle/_ref not built and /root/reference absent")
    return oracle.Oracle("reference")
