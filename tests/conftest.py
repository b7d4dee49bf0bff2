import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the suite vouches for the PRODUCT library: a leaked MPX_LIB_PATH (tools/with_lib.py's probe builds, possibly timing-only) would make
    # every parity test speak for another binary
    if os.environ.get("MPX_LIB_PATH"):
        raise pytest.UsageError("MPX_LIB_PATH=%s is set: the tests run against the in-tree libmpx.so only; unset it" % os.environ["MPX_LIB_PATH"])


@pytest.fixture(scope="session")
def mpx_lib():
    """The built C-ABI library (compiled in-tree by __graft_entry__.build() if missing)."""
    import __graft_entry__ as g
    g.build()
    from network_interpretation_imagenet_amd import _lib
    return _lib.load()


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session", autouse=True)
def _cached_synthetic_state_dicts():
    """synth.make_state_dict(arch) draws 44 M ResNet-101 weights in ~2 s and a hundred tests ask for the same dict: hand out one set of
    tensors per (arch, seed) in a fresh OrderedDict (the tests treat the tensors as read-only; they copy before they change one)."""
    from network_interpretation_imagenet_amd import synth
    real, cache = synth.make_state_dict, {}

    def cached(arch, seed=7):
        key = (arch, seed)
        if key not in cache:
            cache[key] = real(arch, seed)
        return type(cache[key])(cache[key])

    synth.make_state_dict = cached
    yield
    synth.make_state_dict = real
