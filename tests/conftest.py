"""pytest plumbing: registers the `gpu` marker and puts the oracle / host packages on sys.path.

`-m "not gpu"` runs everywhere (no GPU): oracle vs the golden vectors, host logic, C-ABI symbol checks,
gloo world_size-2 sharding.  `-m gpu` tests are the parity tests proper and call the HIP path through
the C-ABI (libmirt.so); they fail loudly if the library or a GPU is missing.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("oracle", "cpp-raytracer-rasterizer_amd"):
    p = os.path.join(ROOT, sub)
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from mirt_oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "survey_appendix_c.json")) as f:
        return json.load(f)
