import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def layouts():
    with open(os.path.join(ROOT, "wfcrl-env_amd", "environments", "layouts.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def kat1():
    with open(os.path.join(ROOT, "tests", "golden", "kat1_demo_notebook.json")) as f:
        return json.load(f)


def gpu_available() -> bool:
    import torch

    return torch.cuda.is_available()
