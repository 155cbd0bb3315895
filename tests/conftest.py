import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/libdfe_oracle.so); built on demand with the committed Makefile."""
    from tests import oracle as orc

    return orc


@pytest.fixture(scope="session")
def dfe():
    """The product package over libdfe.so; needs a GPU for anything but the scalar codec."""
    import depth_estimation_amd as d

    return d


@pytest.fixture(scope="session")
def cuda():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("a test marked gpu ran without a GPU (torch.cuda.is_available() is False)")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _release_dfe_options():
    """Behaviour switches a test forced through dfe_set_option do not leak into the next test."""
    yield
    ctxmod = sys.modules.get("depth-estimation_amd.context") or sys.modules.get("depth_estimation_amd.context")
    if ctxmod is None:
        return
    from depth_estimation_amd._lib import OPTION_KEYS

    for c in list(ctxmod._ctxs.values()):
        if c.handle:
            for k in OPTION_KEYS:
                c.set_option(k, -1 if k != "graphs" else 0)
