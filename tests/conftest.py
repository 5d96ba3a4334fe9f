import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cfmm-routing-code_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


# Some GPU tests emulate an N-GPU job with N contexts (N streams) on ONE device whose kernels WAIT for each other (the
# one-shot all-reduce).  HIP multiplexes streams onto a small pool of hardware queues (4 by default); two such streams on
# one queue would serialise -- rank A's exchange kernel spinning in front of the rank-B kernel it waits for -- which cannot
# happen in the real deployment (one process, one stream, per GPU).  Give the test process more queues than it has ranks.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import c_oracle
    c_oracle.build()
    return c_oracle
