import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]
    return get


def blocks_from_fixture(fx, use_tag):
    """(ui, ut, it) COO blocks in the form oracle.adj.block_adjacency / tagrec_amd.graph take."""
    nu, ni, nt = int(fx["n_user"]), int(fx["n_item"]), int(fx["n_tag"])
    ui = (fx["ui_row"], fx["ui_col"], np.ones(len(fx["ui_row"]), np.float32), (nu, ni))
    if not use_tag:
        return ui, None, None
    ut = (fx["ut_row"], fx["ut_col"], np.ones(len(fx["ut_row"]), np.float32), (nu, nt))
    it = (fx["it_row"], fx["it_col"], np.ones(len(fx["it_row"]), np.float32), (ni, nt))
    return ui, ut, it
