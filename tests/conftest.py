import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "perf: throughput floors on an MI355X (box-dependent; NOT part of the -m gpu parity gate)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure; builds oracle/libwx_oracle.so on first use)."""
    import wx_oracle
    wx_oracle.build()
    return wx_oracle


@pytest.fixture(scope="session")
def wx():
    import subprocess
    lib = os.path.join(ROOT, "waveletsext.jl_amd", "csrc", "libwaveletsext_hip.so")
    if not os.path.exists(lib):                       # fresh checkout: the .so is not in the history
        subprocess.run(["make", "-C", os.path.dirname(lib), "-j%d" % (os.cpu_count() or 4)], check=True)   # ~12 min on 8 cores (profiles/r06_clean_build.txt)
    import waveletsext_jl_amd
    return waveletsext_jl_amd


@pytest.fixture(scope="session")
def kats():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "reference_kats.json")) as f:
        return json.load(f)
