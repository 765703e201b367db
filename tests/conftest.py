import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def fa():
    import fhe_linformer_amd
    return fhe_linformer_amd


_ENGINES = {}


@pytest.fixture(scope="session")
def engine_factory(fa):
    """Session cache of GPU contexts keyed by preset (context creation builds twiddle tables)."""
    def get(preset, **kw):
        key = (preset, tuple(sorted(kw.items())))
        if key not in _ENGINES:
            _ENGINES[key] = fa.Engine(preset, **kw)
        return _ENGINES[key]
    yield get
    for e in _ENGINES.values():
        e.close()
    _ENGINES.clear()
