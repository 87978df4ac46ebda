import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import flatapi
    if not os.path.exists(flatapi.oracle_path()):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(flatapi.ROOT, "oracle"), "oracle"])
    return flatapi.load_oracle()


def _have_ref():
    import flatapi
    return os.path.exists(flatapi.refshim_path())


@pytest.fixture(scope="session", params=[0, 1], ids=["ref-generic", "ref-avx2"])
def ref(request):
    """The compiled reference (oracle/_ref), generic and AVX2 strategies."""
    import flatapi
    if not _have_ref():
        pytest.skip("oracle/_ref not built (needs /root/reference; run `make -C oracle ref`)")
    return request.param


@pytest.fixture()
def reflib(ref):
    import flatapi
    return flatapi.load_ref(ref)
