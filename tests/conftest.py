import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# the incremental index insert (csrc/map_build.hip: map_insert) is meant for maps of several million points; the tests want it on every
# append, at their sizes (read once per process by the library)
os.environ.setdefault("ICPMI_INSERT_MIN", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_bindings as ob
    ob.load()
    return ob


@pytest.fixture(scope="session")
def small_scene():
    from norlab_icp_mapper_amd import synth
    return synth.make_scene(m=60000, n=6000)


@pytest.fixture(scope="session")
def mid_scene():
    from norlab_icp_mapper_amd import synth
    return synth.make_scene(m=200000, n=20000)
