import gzip
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the built .so is not in the git history: build it when a fresh checkout runs the tests first
    # (hipcc cross-compiles gfx950 without a GPU); __graft_entry__.build() does the same
    lib = os.path.join(ROOT, "quade_amd", "lib", "libquade_hip.so")
    if not os.path.exists(lib):
        import subprocess
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "quade_amd", "csrc")])


@pytest.fixture(scope="session")
def finder_vectors():
    with gzip.open(os.path.join(GOLDEN, "finder_vectors.json.gz"), "rb") as fh:
        return json.loads(fh.read().decode())


@pytest.fixture(scope="session")
def bundled_dir():
    return os.path.join(GOLDEN, "bundled")
