"""pytest configuration: registers the `gpu` marker and puts the repo root on sys.path.

`-m "not gpu"` runs on a CPU-only box (oracle vs golden vectors, host logic, C-ABI symbol checks);
`-m gpu` are the parity tests proper and call the HIP path through the C-ABI on a real MI355X.
Nothing here reads /root/reference.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc

    orc.lib()
    return orc


@pytest.fixture(scope="session")
def kat():
    import json

    with open(os.path.join(ROOT, "tests", "golden", "poseidon_kat.json")) as f:
        return json.load(f)
