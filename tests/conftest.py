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


@pytest.fixture(autouse=True, scope="session")
def _plan_free_entry_stays_on_the_general_kernels():
    """rsp_column_sums_device plans for itself by default (include/rcppsparse_hip.h): after a call or two with the same
    offsets it takes the lean / columns form.  The parity tests of this suite were written about the kernels they name --
    many compare two plan-free calls bit for bit -- so the process default here is "auto_plan" OFF; the entry's own
    planning has its module (tests/test_gpu_autoplan.py switches it on) and bench.py's child processes run with the
    library's default (on)."""
    from rcppsparse_amd import capi
    capi.load()
    capi.set_auto_plan(False)
    yield


def golden_names():
    return sorted(os.path.splitext(os.path.basename(f))[0]
                  for f in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def load_golden(name):
    d = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    return {k: d[k] for k in d.files}


@pytest.fixture(scope="session")
def golden():
    return {n: load_golden(n) for n in golden_names()}
