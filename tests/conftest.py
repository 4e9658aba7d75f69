import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pkg(name=""):
    """import 3dssd_amd[.name] (the directory name is not a Python identifier)."""
    return importlib.import_module("3dssd_amd" + ("." + name if name else ""))


@pytest.fixture(scope="session")
def oracle():
    from oracle import sa_oracle
    sa_oracle.lib()
    return sa_oracle


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    pkg("utils._native").lib()   # fail loudly if the HIP extension is missing
    return torch.device("cuda:0")
