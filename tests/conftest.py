import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
CNR = os.path.join(GOLDEN, "cnr-2000")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Builds the native pieces once per session (hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as ge
    ge.build()


@pytest.fixture(scope="session")
def cnr_oracle():
    from oracle import oracle as O
    g = O.OracleGraph.load(CNR)
    rowptr, succ, arcs = g.scan()
    return g, rowptr, succ


def make_graph(tmp_path_factory, name, n, m, seed, p_copy=0.5, **store_kw):
    """Synthetic graph -> BV files; returns (basename, rowptr, succ)."""
    from webgraph_amd import tools as T
    d = tmp_path_factory.mktemp(name)
    base = str(d / name)
    rowptr, succ = T.generate(n, m, seed=seed, p_copy=p_copy)
    T.store(base, rowptr, succ, **store_kw)
    return base, rowptr, succ
