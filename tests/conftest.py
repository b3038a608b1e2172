import json
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def load_golden(name):
    with open(os.path.join(HERE, "golden", name), encoding="utf-8") as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    from _oracle import Oracle
    return Oracle()
