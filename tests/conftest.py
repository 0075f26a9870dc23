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


def source_stamp():
    """the stamp csrc/Makefile embeds in libicpmi.so (icpmi_build_info): SHA-256 over the kernel sources in the Makefile's order"""
    import hashlib
    d = os.path.join(ROOT, "norlab_icp_mapper_amd", "csrc")
    names = sorted(f for f in os.listdir(d) if f.endswith(".hip"))
    h = hashlib.sha256()
    for p in [os.path.join(d, n) for n in names] + [os.path.join(d, "common.h"), os.path.join(d, "solve.h"), os.path.join(ROOT, "include", "icpmi.h")]:
        h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def pytest_collection_modifyitems(config, items):
    """GPU tests run the prebuilt libicpmi.so that travelled with the tree (git-ignored): refuse one that was built from other sources"""
    if not any(i.get_closest_marker("gpu") for i in items):
        return
    try:
        import torch
        if not torch.cuda.is_available():
            return
    except Exception:
        return
    from norlab_icp_mapper_amd import _capi
    info = _capi.load().icpmi_build_info().decode()
    want = source_stamp()
    if not info.endswith("src:" + want):
        raise pytest.UsageError(f"libicpmi.so reports '{info}', the tree's kernel sources hash to {want}: rebuild (python __graft_entry__.py) before running the GPU tests")


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
