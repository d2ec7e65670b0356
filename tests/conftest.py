import os
import sys

import pytest

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, _HERE)
sys.path.insert(0, os.path.join(os.path.dirname(_HERE), "oracle"))
sys.path.insert(0, os.path.dirname(_HERE))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: larger CPU cases")
    config.addinivalue_line("markers", "perf: wall-time regression bounds on the MI355X (not part of -m gpu: a slow box must not fail correctness)")


@pytest.fixture(scope="session")
def orc():
    from helpers import oracle

    return oracle()
