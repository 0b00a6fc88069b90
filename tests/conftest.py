import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The built libraries are git-ignored: in a fresh checkout build them before the first test imports them
    (hipcc cross-compiles gfx950 without a GPU; a few minutes once, nothing when they are up to date)."""
    import subprocess

    missing = [d for d, so in (("orbit_amd/csrc", "orbit_amd/liborbit_cull.so"), ("orbit_amd/host", "orbit_amd/liborbit_host.so"),
                               ("examples", "examples/native_frame"))
               if not os.path.exists(os.path.join(ROOT, so))]
    for d in missing:
        subprocess.run(["make", "-C", os.path.join(ROOT, d), "-j8", "-s"], check=True)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o

    o.build()
    o.lib()
    return o
