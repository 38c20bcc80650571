import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(HERE, "golden")
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def gold_json(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def gold_npz(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


@pytest.fixture(scope="session")
def lib():
    """libpba.so, built in-tree if needed.  Never falls back to anything else."""
    from pacbioassembly_amd import build, _lib
    build.build()
    return _lib.load()


@pytest.fixture(scope="session")
def oracle():
    from oraclelib import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def ctx(lib):
    """A device context.  On a box without a gfx950 GPU this FAILS (it does not skip): the GPU
    tests must never pass on a fallback."""
    from pacbioassembly_amd import Context
    c = Context(0)
    yield c
    c.close()


MASK_PAT = "111*11*11*1*1111"
