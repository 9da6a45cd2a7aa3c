"""rcppsparse_amd -- MI355X-native columnSums for RcppSparse's dgCMatrix hot path.

The product is the C ABI in ``include/rcppsparse_hip.h`` (``librcppsparse_hip.so``,
hand-written HIP for gfx950) plus the C++ host mirror of the reference interface
under ``host/``.  The Python modules here are plumbing around it:

* ``capi``     -- ctypes binding of the C ABI (tests, bench, multi-GPU driver)
* ``synth``    -- synthetic rsparsematrix-like inputs
* ``sharded``  -- column-range data parallelism: one process per GPU, RCCL gatherv
* ``hostseam`` -- ctypes binding of the Rcpp-free build of the host mirror
"""
from . import capi, synth  # noqa: F401

__all__ = ["capi", "synth"]
__version__ = "0.1.0"
