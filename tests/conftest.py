import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """The .so files are build artefacts (git-ignored): build them when a fresh checkout runs the tests first."""
    lib = os.path.join(ROOT, "fhe-linformer_amd", "libfhelin_amd.so")
    drv = os.path.join(ROOT, "tests", "shim", "shim_forward")
    if not (os.path.exists(lib) and os.path.exists(drv)):
        import __graft_entry__
        __graft_entry__.build()
    yield


@pytest.fixture(scope="session")
def orc():
    import oracle
    oracle.lib()
    # OpenMP sees every core of the host, the cgroup grants fewer: dynamic schedules over more threads than CPUs crawl
    cpus = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cpus = min(cpus, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    oracle.set_threads(min(cpus, 16))
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
