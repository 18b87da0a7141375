import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def load_golden(name):
    import numpy as np
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def split_tables(g):
    """tables_cat -> list of (n_k, d) fp32 arrays."""
    import numpy as np
    d = int(g["m_spa"])
    out, p = [], 0
    for n in g["ln_emb"]:
        out.append(np.ascontiguousarray(g["tables_cat"][p:p + int(n) * d].reshape(int(n), d)))
        p += int(n) * d
    return out


def split_indices(g):
    """-> (lS_o (T,B) int64, list of T int64 index arrays)"""
    import numpy as np
    if "lS_i_stacked" in g.files:
        return g["lS_o"], [np.ascontiguousarray(r) for r in g["lS_i_stacked"]]
    out, p = [], 0
    for n in g["lS_i_nnz"]:
        out.append(np.ascontiguousarray(g["lS_i_cat"][p:p + int(n)]))
        p += int(n)
    return g["lS_o"], out


def split_weights(g):
    import numpy as np
    if "vW_cat" not in g.files:
        return None
    out, p = [], 0
    for n in g["ln_emb"]:
        out.append(np.ascontiguousarray(g["vW_cat"][p:p + int(n)]))
        p += int(n)
    return out
