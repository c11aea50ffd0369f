import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def kat():
    with open(os.path.join(ROOT, "tests", "golden", "reference_kat.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="session", autouse=True)
def _native_artifacts_built():
    """The HIP library, the synth harness, the C++ host shim/binary and the oracle are built in-tree
    (they travel with the snapshot).  If any of them is missing or older than its sources -- e.g. on a
    fresh checkout -- build it now (hipcc cross-compiles without a GPU; about a minute)."""
    from fqtk_amd import build as fb
    from oracle import oracle as O
    fb.build(force=False, verbose=False)
    O.build()
