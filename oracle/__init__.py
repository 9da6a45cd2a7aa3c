"""oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of RcppSparse's ``columnSums`` hot path (reference
``src/example.cpp:26-32`` driving ``Matrix::InnerIterator``,
``inst/include/RcppSparse.h:218-233``).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  Nothing under ``rcppsparse_amd/`` imports it.

Parity status: **unpinned by the reference's own tests** (it has none for this
path).  Pinned here by the one reference-derived known
answer (``vignettes/Documentation.Rmd:213-216``) plus a SciPy cross-check of
the restatement; see ``tests/test_oracle.py`` and ``tests/golden/``.

Two restatements are provided:

* ``column_sums`` etc. -- ctypes into ``liboracle.so`` (C, ``-O2``, no
  fast-math), the one used for anything bigger than toy sizes and as the timed
  CPU baseline (1 thread: the reference path has no OpenMP);
* ``column_sums_py`` -- a pure-Python loop for tiny cases, an independent
  second transcription of the same seven lines.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_THREADS_LIB_PATH = os.path.join(_HERE, "liboracle_threads.so")
_lib = None
_threads_lib = None


def _make(target: str, path: str, sources, force: bool) -> str:
    srcs = [os.path.join(_HERE, s) for s in sources]

    def stale():
        return (force or not os.path.exists(path)
                or any(os.path.getmtime(path) < os.path.getmtime(s) for s in srcs))
    if stale():
        import fcntl
        fd = os.open(os.path.join(_HERE, ".build.lock"), os.O_CREAT | os.O_RDWR, 0o644)
        try:
            fcntl.flock(fd, fcntl.LOCK_EX)       # several processes may import at once
            if stale():
                subprocess.run(["make", "-C", _HERE, target], check=True, stdout=subprocess.DEVNULL)
        finally:
            fcntl.flock(fd, fcntl.LOCK_UN)
            os.close(fd)
    return path


def build(force: bool = False) -> str:
    """Compile liboracle.so (and the threaded bench variant) with the committed Makefile
    (gcc, no fast-math)."""
    _make("liboracle_threads.so", _THREADS_LIB_PATH, ["colsums_threads.c", "colsums_oracle.c"], force)
    return _make("liboracle.so", _LIB_PATH, ["colsums_oracle.c"], force)


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        try:
            build()
        except Exception:
            # no compiler / read-only tree (GPU box): use the prebuilt file
            if not os.path.exists(_LIB_PATH):
                raise
        L = ctypes.CDLL(_LIB_PATH)
        dp = ctypes.POINTER(ctypes.c_double)
        ip = ctypes.POINTER(ctypes.c_int32)
        i32, u64 = ctypes.c_int32, ctypes.c_uint64
        L.oracle_column_sums.argtypes = [dp, ip, ip, i32, i32, dp]
        L.oracle_col_sums.argtypes = [dp, ip, i32, dp]
        L.oracle_col_means.argtypes = [dp, ip, i32, i32, dp]
        L.oracle_row_sums.argtypes = [dp, ip, ip, i32, i32, dp]
        L.oracle_row_means.argtypes = [dp, ip, ip, i32, i32, dp]
        L.oracle_column_abs_sums.argtypes = [dp, ip, i32, dp]
        L.oracle_gen_values.argtypes = [dp, u64, u64, u64, ctypes.c_int]
        L.oracle_column_reduce.argtypes = [dp, ip, i32, ctypes.c_int, dp]
        L.oracle_column_reduce.restype = None
        L.oracle_column_sums_in_rows.argtypes = [dp, ip, ip, i32, ctypes.c_void_p, ctypes.c_int, dp]
        L.oracle_column_sums_in_rows.restype = None
        L.oracle_crossprod.argtypes = [dp, ip, ip, i32, dp]
        L.oracle_crossprod.restype = None
        L.oracle_gen_row_indices.argtypes = [ip, ip, i32, i32, i32, u64]
        L.oracle_gen_row_indices.restype = None
        L.oracle_row_sums_accumulate.argtypes = [dp, ip, ctypes.c_int64, dp, dp]
        L.oracle_row_sums_accumulate.restype = None
        L.oracle_gen_value.argtypes = [u64, u64, ctypes.c_int]
        L.oracle_gen_value.restype = ctypes.c_double
        for f in ("oracle_column_sums", "oracle_col_sums", "oracle_col_means",
                  "oracle_row_sums", "oracle_row_means", "oracle_column_abs_sums",
                  "oracle_gen_values"):
            getattr(L, f).restype = None
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _ip(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


def _prep(x, p, i=None):
    x = np.ascontiguousarray(x, dtype=np.float64)
    p = np.ascontiguousarray(p, dtype=np.int32)
    if i is not None:
        i = np.ascontiguousarray(i, dtype=np.int32)
    return x, p, i


def column_sums(x, p, ncol=None, i=None, nrow=0) -> np.ndarray:
    """example.cpp:26-32 via the InnerIterator restatement.  ``i`` is never read."""
    x, p, i = _prep(x, p, i)
    ncol = len(p) - 1 if ncol is None else int(ncol)
    out = np.empty(ncol, dtype=np.float64)
    ii = _ip(i) if i is not None else ctypes.POINTER(ctypes.c_int32)()
    lib().oracle_column_sums(_dp(x), ii, _ip(p), int(nrow), ncol, _dp(out))
    return out


def threads_lib() -> ctypes.CDLL:
    """liboracle_threads.so: the same per-column loop under an OpenMP parallel-for (bench only)."""
    global _threads_lib
    if _threads_lib is None:
        try:
            build()
        except Exception:
            if not os.path.exists(_THREADS_LIB_PATH):
                raise
        L = ctypes.CDLL(_THREADS_LIB_PATH)
        dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32)
        L.oracle_threads_column_sums.argtypes = [dp, ip, ctypes.c_int32, dp, ctypes.c_int]
        L.oracle_threads_column_sums.restype = None
        L.oracle_threads_gen_values.argtypes = [dp, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64,
                                                ctypes.c_int, ctypes.c_int]
        L.oracle_threads_gen_values.restype = None
        L.oracle_threads_max.restype = ctypes.c_int
        _threads_lib = L
    return _threads_lib


def column_sums_threads(x, p, nthreads: int) -> np.ndarray:
    """example.cpp:26-32 with the column loop spread over host threads (bit-identical results)."""
    x, p, _ = _prep(x, p)
    out = np.empty(len(p) - 1, dtype=np.float64)
    threads_lib().oracle_threads_column_sums(_dp(x), _ip(p), len(p) - 1, _dp(out), int(nthreads))
    return out


def gen_values_threads(n, seed, first_idx=0, kind=0, nthreads: int = 1) -> np.ndarray:
    """gen_values filled by several threads (first touch spreads the pages over NUMA nodes)."""
    out = np.empty(int(n), dtype=np.float64)
    threads_lib().oracle_threads_gen_values(_dp(out), int(n), int(seed), int(first_idx), int(kind),
                                            int(nthreads))
    return out


def col_sums(x, p, ncol=None) -> np.ndarray:
    """RcppSparse.h:131-137."""
    x, p, _ = _prep(x, p)
    ncol = len(p) - 1 if ncol is None else int(ncol)
    out = np.empty(ncol, dtype=np.float64)
    lib().oracle_col_sums(_dp(x), _ip(p), ncol, _dp(out))
    return out


def col_means(x, p, nrow, ncol=None) -> np.ndarray:
    """RcppSparse.h:145-150."""
    x, p, _ = _prep(x, p)
    ncol = len(p) - 1 if ncol is None else int(ncol)
    out = np.empty(ncol, dtype=np.float64)
    lib().oracle_col_means(_dp(x), _ip(p), int(nrow), ncol, _dp(out))
    return out


def row_sums(x, i, p, nrow, ncol=None) -> np.ndarray:
    """RcppSparse.h:138-144."""
    x, p, i = _prep(x, p, i)
    ncol = len(p) - 1 if ncol is None else int(ncol)
    out = np.empty(int(nrow), dtype=np.float64)
    lib().oracle_row_sums(_dp(x), _ip(i), _ip(p), int(nrow), ncol, _dp(out))
    return out


def row_sums_accumulate(x, i, sums, abs_sums=None) -> None:
    """sums[i[j]] += x[j] over a further slab of stored entries (RcppSparse.h:141-143 continued in storage order);
    abs_sums likewise with |x[j]| (the tolerance's scale)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    i = np.ascontiguousarray(i, dtype=np.int32)
    assert sums.dtype == np.float64 and sums.flags.c_contiguous and x.size == i.size
    lib().oracle_row_sums_accumulate(_dp(x), _ip(i), int(x.size), _dp(sums),
                                     _dp(abs_sums) if abs_sums is not None else ctypes.POINTER(ctypes.c_double)())


def row_means(x, i, p, nrow, ncol=None) -> np.ndarray:
    """RcppSparse.h:151-156."""
    x, p, i = _prep(x, p, i)
    ncol = len(p) - 1 if ncol is None else int(ncol)
    out = np.empty(int(nrow), dtype=np.float64)
    lib().oracle_row_means(_dp(x), _ip(i), _ip(p), int(nrow), ncol, _dp(out))
    return out


def column_abs_sums(x, p, ncol=None) -> np.ndarray:
    """Per-column 1-norm: the scale of the 1e-12 tolerance."""
    x, p, _ = _prep(x, p)
    ncol = len(p) - 1 if ncol is None else int(ncol)
    out = np.empty(ncol, dtype=np.float64)
    lib().oracle_column_abs_sums(_dp(x), _ip(p), ncol, _dp(out))
    return out


def column_sums_py(x, p, ncol=None):
    """Pure-Python second transcription of example.cpp:26-32 (tiny inputs only)."""
    ncol = len(p) - 1 if ncol is None else int(ncol)
    sums = [0.0] * ncol                      # NumericVector sums(A.cols())
    for col in range(ncol):                  # for (col = 0; col < A.cols(); ++col)
        index, max_index = int(p[col]), int(p[col + 1])   # InnerIterator ctor
        while index < max_index:             # operator bool
            sums[col] += float(x[index])     # sums(col) += it.value()
            index += 1                       # ++it
    return np.array(sums, dtype=np.float64)


def gen_values(n, seed, first_idx=0, kind=0) -> np.ndarray:
    """Counter-based synthetic x[]; bit-identical to the device generator."""
    out = np.empty(int(n), dtype=np.float64)
    lib().oracle_gen_values(_dp(out), int(n), int(seed), int(first_idx), int(kind))
    return out


def gen_row_indices(p, nrow, seed, c_first=0, c_last=None) -> np.ndarray:
    """Row indices of columns [c_first, c_last); bit-identical to the device generator."""
    p = np.ascontiguousarray(p, dtype=np.int32)
    c_last = len(p) - 1 if c_last is None else int(c_last)
    out = np.empty(int(p[c_last] - p[c_first]), dtype=np.int32)
    lib().oracle_gen_row_indices(_ip(out), _ip(p), int(nrow), int(c_first), c_last, int(seed))
    return out


def column_reduce(x, p, op, ncol=None) -> np.ndarray:
    """InnerIterator loop with body acc += f(value): op 0 v, 1 v*v, 2 |v|."""
    x, p, _ = _prep(x, p)
    ncol = len(p) - 1 if ncol is None else int(ncol)
    out = np.empty(ncol, dtype=np.float64)
    lib().oracle_column_reduce(_dp(x), _ip(p), ncol, int(op), _dp(out))
    return out


def column_sums_in_rows(x, i, p, bitmap, complement=False, ncol=None) -> np.ndarray:
    """Column sums over entries whose row is (not) in the set (restricted iterators, RcppSparse.h:238-321)."""
    x, p, i = _prep(x, p, i)
    bitmap = np.ascontiguousarray(bitmap, dtype=np.uint32)
    ncol = len(p) - 1 if ncol is None else int(ncol)
    out = np.empty(ncol, dtype=np.float64)
    lib().oracle_column_sums_in_rows(_dp(x), _ip(i), _ip(p), ncol, bitmap.ctypes.data, int(bool(complement)),
                                     _dp(out))
    return out


def crossprod(x, i, p, ncol=None) -> np.ndarray:
    """RcppSparse.h:159-194: dense t(A) %*% A by pairwise sorted merges (reference order)."""
    x, p, i = _prep(x, p, i)
    ncol = len(p) - 1 if ncol is None else int(ncol)
    out = np.empty((ncol, ncol), dtype=np.float64, order="F")
    lib().oracle_crossprod(_dp(x), _ip(i), _ip(p), ncol, _dp(out))
    return out
