import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")

# The Rcpp layer's columnSums() answers on the host when the machine has no GPU at all (columnsums_impl.hpp).
# Nothing this suite checks on a GPU box may ever come from that loop: a GPU is REQUIRED in every process the
# tests start, and the tests of the host answer itself take the requirement away explicitly.
os.environ["RCPPSPARSE_REQUIRE_GPU"] = "1"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# The suite runs under the LIBRARY'S DEFAULTS (round 6): rsp_column_sums_device plans for itself
# (include/rcppsparse_hip.h) exactly as it does for a caller who sets nothing.  Tests that are about the general kernels
# pin capi.set_auto_plan(False) themselves (the `launch_mode` fixture's "general" leg, tools/edge_sweep.py), and put the
# default back.  Nothing here loads the library: a pure-CPU test (oracle, host mirror, mock Rcpp) must not need it.


def golden_names():
    return sorted(os.path.splitext(os.path.basename(f))[0]
                  for f in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def load_golden(name):
    d = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    return {k: d[k] for k in d.files}


@pytest.fixture(scope="session")
def golden():
    return {n: load_golden(n) for n in golden_names()}
