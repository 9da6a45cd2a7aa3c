"""ctypes binding of librcppsparse_host.so: the Rcpp-free build of the C++ host
mirror (``host/rcppsparse_core.hpp`` + ``host/columnsums_impl.hpp``).  Test
plumbing: lets Python drive ``RcppSparse::Matrix``'s members and the exported
``columnSums(Matrix&)`` without R."""
from __future__ import annotations

import ctypes
import os

import numpy as np

from . import _build, capi

_lib = None


class SeamError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        capi.load()                       # the seam links librcppsparse_hip.so
        try:
            _build.build_host_seam()
        except Exception:
            if not os.path.exists(_build.HOST_SEAM_PATH):
                raise
        L = ctypes.CDLL(_build.HOST_SEAM_PATH)
        L.seam_last_error.restype = ctypes.c_char_p
        _lib = L
    return _lib


def _slots(m):
    x = np.ascontiguousarray(m["x"], dtype=np.float64)
    i = np.ascontiguousarray(m["i"], dtype=np.int32)
    p = np.ascontiguousarray(m["p"], dtype=np.int32)
    d = np.ascontiguousarray(m["Dim"], dtype=np.int32)
    return x, i, p, d


def _args(x, i, p, d):
    c = ctypes
    return (x.ctypes.data_as(c.c_void_p), i.ctypes.data_as(c.c_void_p),
            p.ctypes.data_as(c.c_void_p), d.ctypes.data_as(c.c_void_p), c.c_int(x.size))


def _check(rc):
    if rc != 0:
        raise SeamError(load().seam_last_error().decode())


def columnSums(m) -> np.ndarray:
    """Exported columnSums(RcppSparse::Matrix&) -> NumericVector, through the HIP shim."""
    x, i, p, d = _slots(m)
    out = np.empty(int(d[1]), dtype=np.float64)
    _check(load().seam_columnSums(*_args(x, i, p, d), out.ctypes.data_as(ctypes.c_void_p)))
    return out


def columnSums_opt(m, require_gpu: int = -1) -> np.ndarray:
    """The same exported function with the R option twin spelled out: 1 = a GPU is required, 0 = the host loop
    may answer when the machine has no GPU, -1 = RCPPSPARSE_REQUIRE_GPU in the environment decides."""
    x, i, p, d = _slots(m)
    out = np.empty(int(d[1]), dtype=np.float64)
    _check(load().seam_columnSums_opt(*_args(x, i, p, d), ctypes.c_int(int(require_gpu)),
                                      out.ctypes.data_as(ctypes.c_void_p)))
    return out


def columnSums_opt2(m, require_gpu: int = -1, min_nnz: int = -1) -> np.ndarray:
    """... and with the offload threshold's option twin: min_nnz >= 0 as options(RcppSparse.min_nnz = n) (matrices with
    fewer stored entries are answered by the host loop although a GPU is present), -1 = RCPPSPARSE_MIN_NNZ / default."""
    x, i, p, d = _slots(m)
    out = np.empty(int(d[1]), dtype=np.float64)
    _check(load().seam_columnSums_opt2(*_args(x, i, p, d), ctypes.c_int(int(require_gpu)), ctypes.c_longlong(int(min_nnz)),
                                       out.ctypes.data_as(ctypes.c_void_p)))
    return out


BACKENDS = ("none", "hip", "cpu")


def backend_for(nnz: int, require_gpu: int = -1, min_nnz: int = -1) -> str:
    """The path columnSums would take now on a matrix of `nnz` stored entries."""
    L = load()
    L.seam_backend_for.argtypes = [ctypes.c_longlong, ctypes.c_int, ctypes.c_longlong]
    return BACKENDS[int(L.seam_backend_for(int(nnz), int(require_gpu), int(min_nnz)))]


def min_nnz(option: int = -1) -> int:
    L = load()
    L.seam_min_nnz.argtypes = [ctypes.c_longlong]
    L.seam_min_nnz.restype = ctypes.c_longlong
    return int(L.seam_min_nnz(int(option)))


def backend(last: bool = False, require_gpu: int = -1) -> str:
    """Which path answers columnSums: now (last=False) or in the most recent call (last=True)."""
    return BACKENDS[int(load().seam_backend(int(bool(last)), int(require_gpu)))]


def columnSums_by_iterator(m) -> np.ndarray:
    x, i, p, d = _slots(m)
    out = np.empty(int(d[1]), dtype=np.float64)
    _check(load().seam_columnSums_by_iterator(*_args(x, i, p, d), out.ctypes.data_as(ctypes.c_void_p)))
    return out


def sizes(m, col0=0):
    x, i, p, d = _slots(m)
    out = (ctypes.c_uint * 6)()
    _check(load().seam_sizes(*_args(x, i, p, d), ctypes.c_int(col0), out))
    return dict(zip(("rows", "cols", "nrow", "ncol", "n_nonzero", "InnerNNZs"), list(out)))


def walk_column(m, col):
    x, i, p, d = _slots(m)
    cap = max(1, int(p[col + 1] - p[col]))
    rows = np.empty(cap, dtype=np.int32)
    vals = np.empty(cap, dtype=np.float64)
    cols = np.empty(cap, dtype=np.int32)
    n = ctypes.c_int(0)
    vp = ctypes.c_void_p
    _check(load().seam_walk_column(*_args(x, i, p, d), ctypes.c_int(col), rows.ctypes.data_as(vp),
                                   vals.ctypes.data_as(vp), cols.ctypes.data_as(vp), ctypes.byref(n)))
    return rows[:n.value], vals[:n.value], cols[:n.value]


def walk_restricted(m, col, s, mode):
    """mode 'in' / 'not_in' (column cursors restricted by row set s) or 'row' (row cursor)."""
    x, i, p, d = _slots(m)
    s = np.ascontiguousarray(s, dtype=np.uint32)
    cap = max(1, int(max(d[0], d[1])))
    idx = np.empty(cap, dtype=np.int32)
    vals = np.empty(cap, dtype=np.float64)
    n = ctypes.c_int(0)
    vp = ctypes.c_void_p
    code = {"in": 0, "not_in": 1, "row": 2}[mode]
    _check(load().seam_walk_restricted(*_args(x, i, p, d), ctypes.c_int(col), s.ctypes.data_as(vp),
                                       ctypes.c_int(s.size), ctypes.c_int(code),
                                       idx.ctypes.data_as(vp), vals.ctypes.data_as(vp), ctypes.byref(n)))
    return idx[:n.value], vals[:n.value]


_DENSE = {"colSums": 0, "rowSums": 1, "colMeans": 2, "rowMeans": 3, "col": 4, "row": 5,
          "crossprod": 6, "at": 7}


def dense(m, which, a=0, b=0) -> np.ndarray:
    x, i, p, d = _slots(m)
    nr, nc = int(d[0]), int(d[1])
    n = {"colSums": nc, "rowSums": nr, "colMeans": nc, "rowMeans": nr, "col": nr, "row": nc,
         "crossprod": nc * nc, "at": 1}[which]
    out = np.empty(max(n, 1), dtype=np.float64)
    _check(load().seam_dense(*_args(x, i, p, d), ctypes.c_int(_DENSE[which]), ctypes.c_int(a),
                             ctypes.c_int(b), out.ctypes.data_as(ctypes.c_void_p)))
    out = out[:n]
    return out.reshape(nc, nc).T if which == "crossprod" else out


def subview(m, which, rows=(), cols=(), a0=0) -> np.ndarray:
    """Sub-view clones / index helpers of the class (see seam_subviews in host_seam.cpp)."""
    x, i, p, d = _slots(m)
    rows = np.ascontiguousarray(rows, dtype=np.int32)
    cols = np.ascontiguousarray(cols, dtype=np.int32)
    nr, nc = int(d[0]), int(d[1])
    sel = {"row_cols": 0, "rows_col": 1, "rows_cols": 2, "cols": 3, "rows": 4, "linear": 5,
           "InnerIndices": 6, "emptyInnerIndices": 7}[which]
    cap = max(rows.size * max(cols.size, nc, 1), nr * max(cols.size, 1), nr + 1, 4)
    out = np.zeros(cap + 2, dtype=np.float64)
    vp = ctypes.c_void_p
    _check(load().seam_subviews(*_args(x, i, p, d), ctypes.c_int(sel), rows.ctypes.data_as(vp),
                                ctypes.c_int(rows.size), cols.ctypes.data_as(vp), ctypes.c_int(cols.size),
                                ctypes.c_int(int(a0)), out.ctypes.data_as(vp)))
    if sel == 0:
        return out[:cols.size]
    if sel == 1:
        return out[:rows.size]
    if sel == 2:
        return out[:rows.size * cols.size].reshape(cols.size, rows.size).T
    if sel == 3:
        return out[:nr * cols.size].reshape(cols.size, nr).T
    if sel == 4:
        return out[:rows.size * nc].reshape(nc, rows.size).T
    if sel == 5:
        return out[:1]
    n = int(out[0])
    return out[1:1 + n].astype(np.int64)


def transpose(m):
    x, i, p, d = _slots(m)
    tx = np.empty(max(x.size, 1), dtype=np.float64)
    ti = np.empty(max(x.size, 1), dtype=np.int32)
    tp = np.empty(int(d[0]) + 1, dtype=np.int32)
    vp = ctypes.c_void_p
    _check(load().seam_transpose(*_args(x, i, p, d), tx.ctypes.data_as(vp), ti.ctypes.data_as(vp),
                                 tp.ctypes.data_as(vp)))
    return {"x": tx[:x.size], "i": ti[:x.size], "p": tp, "Dim": np.array([d[1], d[0]], dtype=np.int32)}


def is_appx_symmetric(m) -> bool:
    x, i, p, d = _slots(m)
    out = ctypes.c_int(0)
    _check(load().seam_is_appx_symmetric(*_args(x, i, p, d), ctypes.byref(out)))
    return bool(out.value)


def construct_from_s4(mask: int) -> None:
    _check(load().seam_construct_from_s4(ctypes.c_int(mask)))


def shares_storage() -> bool:
    return bool(load().seam_shares_storage())
