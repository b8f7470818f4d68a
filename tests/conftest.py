import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import orc
    orc.lib()
    return orc


@pytest.fixture(scope="session")
def seq752():
    """3 seeded 752x480 stereo frames + warps (shared by several tests)."""
    from hybvio_amd import synth
    return synth.stereo_sequence(7, 752, 480, 3)
