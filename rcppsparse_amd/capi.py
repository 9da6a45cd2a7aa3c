"""ctypes binding of ``include/rcppsparse_hip.h`` (librcppsparse_hip.so).

This is plumbing for tests, ``bench.py`` and the multi-GPU driver: the product
is the C ABI itself (what an Rcpp ``columnSums`` would call, see
INTEGRATION.md).  torch is used only as the owner of device memory and streams;
no torch type crosses the ABI -- tensors are passed as raw ``data_ptr()``.

There is no CPU fallback here: if the library is missing or no HIP device is
usable, calls raise ``RspError``.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

from . import _build

RSP_OK = 0
RSP_ERR_NO_DEVICE = 1
RSP_ERR_BAD_ARG = 2
RSP_ERR_HIP = 3
RSP_ERR_WORKSPACE = 4
RSP_ERR_RCCL = 5
RSP_ERR_ALLOC = 6
UNIQUE_ID_BYTES = 128

# every symbol include/rcppsparse_hip.h declares (checked by tests/test_capi_nogpu.py)
EXPORTED_SYMBOLS = (
    "rsp_version", "rsp_last_error", "rsp_device_count",
    "rsp_column_sums_host", "rsp_column_sums_host_multi", "rsp_release_cached",
    "rsp_mcsc_upload", "rsp_mcsc_column_sums", "rsp_mcsc_free",
    "rsp_mcsc_upload_csc", "rsp_mcsc_column_means", "rsp_mcsc_row_sums", "rsp_mcsc_row_means",
    "rsp_csc_upload", "rsp_csc_column_sums", "rsp_csc_column_means", "rsp_csc_free",
    "rsp_column_sums_workspace_bytes", "rsp_column_sums_device", "rsp_column_means_device",
    "rsp_column_sums_device_timed", "rsp_column_reduce_device", "rsp_column_sums_in_rows_device",
    "rsp_column_sums_plan_create", "rsp_column_sums_plan_create_device", "rsp_column_sums_plan_info",
    "rsp_column_sums_planned_device", "rsp_column_sums_plan_destroy", "rsp_column_sums_device_form",
    "rsp_column_sums_in_rows_workspace_bytes", "rsp_column_sums_in_rows_form",
    "rsp_csc_row_form", "rsp_debug_set", "rsp_debug_get",
    "rsp_csc_crossprod", "rsp_crossprod_workspace_bytes", "rsp_crossprod_device",
    "rsp_csc_row_sums", "rsp_csc_row_means", "rsp_row_sums_workspace_bytes", "rsp_row_sums_device",
    "rsp_row_means_device",
    "rsp_partition_columns", "rsp_rebase_offsets",
    "rsp_comm_unique_id", "rsp_comm_init", "rsp_comm_gatherv", "rsp_comm_destroy",
    "rsp_comm_reduce_rows_workspace_bytes", "rsp_comm_reduce_rows", "rsp_add_partials_device",
    "rsp_gen_values_device", "rsp_gen_row_indices_device", "rsp_plan_describe", "rsp_set_crossprod_exact",
    "rsp_debug_read_ceiling_device",
    "rsp_column_sums_plan_ready", "rsp_column_sums_plan_wait", "rsp_debug_plan_image",
    "rsp_shared_result_alloc", "rsp_shared_result_open", "rsp_shared_result_close", "rsp_shared_result_read",
    "rsp_host_barrier_create", "rsp_host_barrier_wait", "rsp_host_barrier_destroy",
    "rsp_crossprod_form", "rsp_debug_exclusive_scan_device",
    "rsp_csc_dims", "rsp_csc_column_form", "rsp_csc_set_planned", "rsp_mcsc_dims", "rsp_mcsc_shard_info",
    "rsp_mcsc_set_gather", "rsp_mcsc_set_launch", "rsp_mcsc_config", "rsp_mcsc_result_buffer", "rsp_mcsc_wrap_device",
    "rsp_mcsc_last_call_stamps", "rsp_mcsc_shard_kernel_ms", "rsp_rccl_info",
    "rsp_column_sums_device_settle",
    "rsp_shared_host_open", "rsp_shared_host_close", "rsp_copy_to_host_async", "rsp_device_can_access_peer",
)
GATHER_MODES = {"d2h": 0, "rccl": 1, "stores": 2, "none": 3, "blit": 4}
LAUNCH_MODES = {"serial": 0, "workers": 1}


class RspError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"rcppsparse_hip error {code}: {msg}")
        self.code = code


_lib = None


def load(build: bool = True) -> ctypes.CDLL:
    """dlopen librcppsparse_hip.so (building it first when hipcc is present)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own libamdhip64 / librccl with the same sonames; importing it
    # first makes this library share that one HIP runtime (so tensor pointers and
    # streams are valid on both sides).
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional for pure-host use
        pass
    path = _build.LIB_PATH
    if os.environ.get("RSP_AB_LIB"):          # A/B measurements: a saved build of the library instead of the tree's
        path, build = os.environ["RSP_AB_LIB"], False
    if build and _build.have_hipcc():
        try:
            _build.build_library()
        except Exception:
            if not os.path.exists(path):
                raise
    if not os.path.exists(path):
        raise RspError(RSP_ERR_NO_DEVICE, f"{path} is missing and cannot be built (no hipcc)")
    L = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    c = ctypes
    dp, ip = c.POINTER(c.c_double), c.POINTER(c.c_int32)
    vp, i32, i64, u64 = c.c_void_p, c.c_int32, c.c_int64, c.c_uint64
    L.rsp_version.restype = c.c_char_p
    L.rsp_last_error.restype = c.c_char_p
    L.rsp_device_count.argtypes = [c.POINTER(c.c_int)]
    L.rsp_column_sums_host.argtypes = [dp, ip, i32, i64, dp, c.c_int]
    L.rsp_column_sums_host_multi.argtypes = [dp, ip, i32, i64, dp, c.POINTER(c.c_int), c.c_int]
    L.rsp_mcsc_upload.argtypes = [dp, ip, i32, i32, i64, c.POINTER(c.c_int), c.c_int, c.POINTER(vp)]
    L.rsp_mcsc_column_sums.argtypes = [vp, dp]
    L.rsp_mcsc_upload_csc.argtypes = [dp, ip, ip, i32, i32, i64, c.POINTER(c.c_int), c.c_int, c.POINTER(vp)]
    L.rsp_mcsc_column_means.argtypes = [vp, dp]
    L.rsp_mcsc_row_sums.argtypes = [vp, dp]
    L.rsp_mcsc_row_means.argtypes = [vp, dp]
    L.rsp_mcsc_free.argtypes = [vp]
    L.rsp_mcsc_set_gather.argtypes = [vp, c.c_int]
    L.rsp_mcsc_set_launch.argtypes = [vp, c.c_int]
    L.rsp_mcsc_config.argtypes = [vp, ip]
    L.rsp_mcsc_result_buffer.argtypes = [vp]
    L.rsp_mcsc_result_buffer.restype = dp
    L.rsp_mcsc_wrap_device.argtypes = [c.c_int, c.POINTER(c.c_int), c.POINTER(vp), c.POINTER(vp), c.POINTER(vp), ip,
                                       c.POINTER(i64), i32, c.POINTER(vp)]
    L.rsp_mcsc_last_call_stamps.argtypes = [vp, dp, c.c_int]
    L.rsp_mcsc_shard_kernel_ms.argtypes = [vp, i32, c.c_int, c.POINTER(c.c_float)]
    L.rsp_csc_upload.argtypes = [dp, ip, ip, i32, i32, i64, c.c_int, c.POINTER(vp)]
    L.rsp_csc_column_sums.argtypes = [vp, dp]
    L.rsp_csc_column_means.argtypes = [vp, dp]
    L.rsp_csc_free.argtypes = [vp]
    L.rsp_column_sums_workspace_bytes.argtypes = [i32, i64]
    L.rsp_column_sums_workspace_bytes.restype = c.c_size_t
    L.rsp_column_sums_device.argtypes = [vp, vp, i32, i64, vp, vp, c.c_size_t, vp]
    L.rsp_column_means_device.argtypes = [vp, vp, i32, i32, i64, vp, vp, c.c_size_t, vp]
    L.rsp_column_reduce_device.argtypes = [vp, vp, i32, i64, c.c_int, vp, vp, c.c_size_t, vp]
    L.rsp_column_sums_in_rows_device.argtypes = [vp, vp, vp, i32, i32, i64, vp, c.c_int, vp, vp,
                                                 c.c_size_t, vp]
    L.rsp_column_sums_device_timed.argtypes = [vp, vp, i32, i64, vp, vp, c.c_size_t, vp, c.c_int,
                                               c.POINTER(c.c_float)]
    L.rsp_column_sums_plan_create.argtypes = [ip, i32, i64, c.c_int, c.POINTER(vp)]
    L.rsp_column_sums_plan_create_device.argtypes = [vp, i32, i64, vp, c.POINTER(vp)]
    L.rsp_column_sums_plan_info.argtypes = [vp, ip, c.POINTER(c.c_double)]
    L.rsp_column_sums_planned_device.argtypes = [vp, vp, vp, i32, i64, i32, vp, vp, c.c_size_t, vp]
    L.rsp_csc_dims.argtypes = [vp, ip, ip, c.POINTER(i64)]
    L.rsp_csc_column_form.argtypes = [vp]
    L.rsp_csc_set_planned.argtypes = [vp, c.c_int]
    L.rsp_mcsc_dims.argtypes = [vp, ip, ip, ip]
    L.rsp_mcsc_shard_info.argtypes = [vp, i32, ip]
    L.rsp_column_sums_plan_destroy.argtypes = [vp]
    L.rsp_column_sums_plan_ready.argtypes = [vp]
    L.rsp_column_sums_plan_wait.argtypes = [vp]
    L.rsp_debug_plan_image.argtypes = [vp, c.c_int, vp, c.c_size_t, c.POINTER(c.c_size_t)]
    L.rsp_debug_set.argtypes = [c.c_char_p, c.c_int]
    L.rsp_debug_get.argtypes = [c.c_char_p, c.POINTER(c.c_int)]
    L.rsp_column_sums_device_form.argtypes = [vp, i32, i64, c.c_int]
    L.rsp_column_sums_device_settle.argtypes = [vp, i32, i64, vp]
    L.rsp_csc_row_form.argtypes = [c.c_void_p]
    L.rsp_column_sums_in_rows_form.argtypes = [i32, i32, i64, c.c_size_t]
    L.rsp_column_sums_in_rows_workspace_bytes.argtypes = [i32, i32, i64]
    L.rsp_column_sums_in_rows_workspace_bytes.restype = c.c_size_t
    L.rsp_csc_crossprod.argtypes = [vp, dp]
    L.rsp_crossprod_device.argtypes = [vp, vp, vp, i32, i32, i64, vp, vp, c.c_size_t, vp]
    L.rsp_crossprod_form.argtypes = [i32, i32, i64]
    L.rsp_crossprod_workspace_bytes.argtypes = [i32, i32, i64]
    L.rsp_crossprod_workspace_bytes.restype = c.c_size_t
    L.rsp_csc_row_sums.argtypes = [vp, dp]
    L.rsp_csc_row_means.argtypes = [vp, dp]
    L.rsp_row_sums_workspace_bytes.argtypes = [i32, i64]
    L.rsp_row_sums_workspace_bytes.restype = c.c_size_t
    L.rsp_row_sums_device.argtypes = [vp, vp, i32, i64, vp, vp, c.c_size_t, vp]
    L.rsp_row_means_device.argtypes = [vp, vp, i32, i32, i64, vp, vp, c.c_size_t, vp]
    L.rsp_partition_columns.argtypes = [ip, i32, i32, ip]
    L.rsp_rebase_offsets.argtypes = [ip, i32, i32, ip]
    L.rsp_comm_unique_id.argtypes = [vp]
    L.rsp_rccl_info.argtypes = [c.POINTER(c.c_int), c.c_char_p, c.c_size_t]
    L.rsp_comm_init.argtypes = [vp, c.c_int, c.c_int, c.c_int, c.POINTER(vp)]
    L.rsp_comm_gatherv.argtypes = [vp, vp, i64, vp, c.POINTER(i64), c.POINTER(i64), c.c_int, vp]
    L.rsp_comm_destroy.argtypes = [vp]
    L.rsp_comm_reduce_rows_workspace_bytes.argtypes = [c.c_int, i32]
    L.rsp_comm_reduce_rows_workspace_bytes.restype = c.c_size_t
    L.rsp_comm_reduce_rows.argtypes = [vp, vp, i32, i32, vp, vp, c.c_size_t, c.c_int, vp]
    L.rsp_add_partials_device.argtypes = [vp, i32, i64, i64, i32, vp, vp]
    L.rsp_shared_result_alloc.argtypes = [c.c_size_t, c.POINTER(vp), vp]
    L.rsp_shared_result_open.argtypes = [vp, c.POINTER(vp)]
    L.rsp_shared_result_close.argtypes = [vp, c.c_int]
    L.rsp_shared_result_read.argtypes = [vp, c.c_size_t, vp, c.c_size_t, vp]
    L.rsp_shared_host_open.argtypes = [c.c_char_p, c.c_size_t, c.c_int, c.POINTER(vp)]
    L.rsp_shared_host_close.argtypes = [vp, c.c_size_t, c.c_char_p]
    L.rsp_copy_to_host_async.argtypes = [vp, vp, i64, vp]
    L.rsp_device_can_access_peer.argtypes = [c.c_int, c.c_int, c.POINTER(c.c_int)]
    L.rsp_host_barrier_create.argtypes = [c.c_char_p, c.c_int, c.c_int, c.POINTER(vp)]
    L.rsp_host_barrier_wait.argtypes = [vp, c.c_double]
    L.rsp_host_barrier_destroy.argtypes = [vp]
    L.rsp_gen_values_device.argtypes = [vp, i64, u64, u64, c.c_int, vp]
    L.rsp_gen_row_indices_device.argtypes = [vp, vp, i32, i32, u64, vp]
    L.rsp_set_crossprod_exact.argtypes = [c.c_int]
    L.rsp_plan_describe.argtypes = [i64, ip]
    L.rsp_debug_exclusive_scan_device.argtypes = [vp, vp, i64, vp]
    L.rsp_debug_read_ceiling_device.argtypes = [vp, i64, vp, vp, c.c_int, c.POINTER(c.c_float)]
    _lib = L
    return L


def _check(rc: int) -> None:
    if rc != RSP_OK:
        raise RspError(rc, load().rsp_last_error().decode("utf-8", "replace"))


def _dp(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _ip(a):
    if a is None:
        return ctypes.POINTER(ctypes.c_int32)()
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


def version() -> str:
    return load().rsp_version().decode()


def device_count() -> int:
    n = ctypes.c_int(0)
    _check(load().rsp_device_count(ctypes.byref(n)))
    return n.value


def debug_set(key: str, value: int) -> None:
    """One process-wide measurement / test knob (rsp_debug_set; the keys are listed in include/rcppsparse_hip.h)."""
    _check(load().rsp_debug_set(key.encode(), int(value)))


def debug_get(key: str) -> int:
    v = ctypes.c_int(0)
    _check(load().rsp_debug_get(key.encode(), ctypes.byref(v)))
    return int(v.value)


def set_tuning(chunk_rows: int = 0) -> None:
    debug_set("chunk_rows", chunk_rows)


def set_auto_plan(on: bool = True) -> None:
    """rsp_column_sums_device / rsp_column_means_device plan for themselves (default) or stay on the general kernels."""
    debug_set("auto_plan", int(bool(on)))


def column_sums_device_form(p_t, nnz: int, wait: bool = False) -> str:
    """The form plan-free calls on these offsets take now (rsp_column_sums_device_form): 'general', 'lean', 'columns', or
    'unknown' (no call with this key yet, or the inspection's result has not been seen; wait=True blocks for it)."""
    r = int(load().rsp_column_sums_device_form(p_t.data_ptr(), p_t.numel() - 1, int(nnz), int(bool(wait))))
    return {0: "general", 2: "lean", 3: "columns"}.get(r, "unknown")


def column_sums_device_settle(p_t, nnz: int, stream=None) -> str:
    """Makes the plan-free entry's plan for these offsets now (if there is none), waits for the inspection and returns the
    form every later call with this key takes: 'general', 'lean' or 'columns' (rsp_column_sums_device_settle).  From here on
    calls with the key are bit-identical run to run."""
    r = int(load().rsp_column_sums_device_settle(p_t.data_ptr(), p_t.numel() - 1, int(nnz), _stream_ptr(stream)))
    return {0: "general", 2: "lean", 3: "columns"}.get(r, "unknown")


def set_crossprod_exact(exact: bool) -> None:
    """True: crossprod keeps the reference's accumulation order (bit-identical) on every shape; False (default):
    tall matrices (ncol <= 256, long columns) go to the matrix-core form (rsp_set_crossprod_exact)."""
    _check(load().rsp_set_crossprod_exact(int(bool(exact))))


def set_taper(tail_permille: int = -1, tail_chunk_rows: int = -1) -> None:
    """(0, 0) = no taper, (-1, -1) = the library's default."""
    debug_set("taper_permille", tail_permille)
    debug_set("taper_rows", tail_chunk_rows)


def plan_describe(nnz: int) -> dict:
    """How a call over nnz entries would be chunked now (rsp_plan_describe)."""
    out = np.zeros(4, dtype=np.int32)
    _check(load().rsp_plan_describe(int(nnz), _ip(out)))
    return {"body_elems": int(out[0]), "nbody": int(out[1]), "tail_elems": int(out[2]), "nchunks": int(out[3])}


def set_lean(on=True) -> None:
    """Plans made from now on: 0 / False never take the lean form, 1 / True where it is the faster one (default: every
    column <= 64 entries and a mean of at most 60), 2 wherever it applies at all (rsp_debug_set "lean"; A/B measurements, tests)."""
    debug_set("lean", int(on))


def set_columns_form(mode: int = 1) -> None:
    """Plans made from now on: 0 never the columns form, 1 where it is the faster one (default), 2 wherever the
    kernel can run at all (rsp_debug_set "columns_form"; edge measurements)."""
    debug_set("columns_form", int(mode))


def set_experiment(variant: int = 0) -> None:
    debug_set("experiment", int(variant))


# ---------------------------------------------------------------- host paths
def column_sums_host(x, p, ncol=None, device: int = 0) -> np.ndarray:
    """One-shot host->device->host columnSums (what the Rcpp columnSums calls)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    p = np.ascontiguousarray(p, dtype=np.int32)
    ncol = len(p) - 1 if ncol is None else int(ncol)
    out = np.empty(ncol, dtype=np.float64)
    _check(load().rsp_column_sums_host(_dp(x), _ip(p), ncol, x.size, _dp(out), device))
    return out


def release_cached() -> None:
    """Frees what the library keeps between calls (the one-shot entry's per-device stream and buffers)."""
    _check(load().rsp_release_cached())


def column_sums_host_multi(x, p, ncol=None, devices=None) -> np.ndarray:
    """One-shot host columnSums over several GPUs (one shard per entry of `devices`;
    None = every visible device)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    p = np.ascontiguousarray(p, dtype=np.int32)
    ncol = len(p) - 1 if ncol is None else int(ncol)
    out = np.empty(ncol, dtype=np.float64)
    if devices is None:
        arr, n = None, 0
    else:
        n = len(devices)
        arr = (ctypes.c_int * n)(*[int(d) for d in devices])
    _check(load().rsp_column_sums_host_multi(_dp(x), _ip(p), ncol, x.size, _dp(out), arr, n))
    return out


class MultiDeviceCSC:
    """dgCMatrix resident on several GPUs of the node (column ranges), one process.  With `i` the row indices
    are kept on the devices too and the row-wise entries work (rsp_mcsc_upload_csc)."""

    def __init__(self, x, p, dim, devices=None, i=None):
        x = np.ascontiguousarray(x, dtype=np.float64)
        p = np.ascontiguousarray(p, dtype=np.int32)
        self.nrow, self.ncol = int(dim[0]), int(dim[1])
        self._h = ctypes.c_void_p()
        if devices is None:
            arr, n = None, 0
        else:
            n = len(devices)
            arr = (ctypes.c_int * n)(*[int(d) for d in devices])
        if i is None:
            _check(load().rsp_mcsc_upload(_dp(x), _ip(p), self.nrow, self.ncol, x.size, arr, n,
                                          ctypes.byref(self._h)))
        else:
            i = np.ascontiguousarray(i, dtype=np.int32)
            _check(load().rsp_mcsc_upload_csc(_dp(x), _ip(i), _ip(p), self.nrow, self.ncol, x.size, arr, n,
                                              ctypes.byref(self._h)))

    @classmethod
    def wrap_device(cls, x_ts, p_ts, nrow: int, i_ts=None):
        """The handle over shards that already live in the devices' HBM (rsp_mcsc_wrap_device): x_ts[k] / p_ts[k] are
        torch tensors on shard k's device (float64 entries, int32 rebased offsets), i_ts optionally the row indices.
        Nothing is copied; the tensors are kept alive by the object."""
        self = cls.__new__(cls)
        G = len(x_ts)
        self._keep = (list(x_ts), list(p_ts), None if i_ts is None else list(i_ts))
        self.nrow = int(nrow)
        ncols = [int(pt.numel()) - 1 for pt in p_ts]
        self.ncol = int(sum(ncols))
        self._h = ctypes.c_void_p()
        vps = ctypes.c_void_p * G
        devs = (ctypes.c_int * G)(*[int(xt.device.index or 0) for xt in x_ts])
        dx = vps(*[xt.data_ptr() for xt in x_ts])
        dpp = vps(*[pt.data_ptr() for pt in p_ts])
        di = None if i_ts is None else vps(*[it.data_ptr() for it in i_ts])
        nc = (ctypes.c_int32 * G)(*ncols)
        nz = (ctypes.c_int64 * G)(*[int(xt.numel()) for xt in x_ts])
        _check(load().rsp_mcsc_wrap_device(G, devs, dx, di, dpp, nc, nz, self.nrow, ctypes.byref(self._h)))
        return self

    def _out(self, fn, n):
        out = np.empty(n, dtype=np.float64)
        _check(fn(self._h, _dp(out)))
        return out

    def column_sums(self, out=None) -> np.ndarray:
        """out: a float64 vector of ncol entries to fill (a fresh one otherwise); `self.result_buffer()` = no host copy."""
        if out is None:
            return self._out(load().rsp_mcsc_column_sums, self.ncol)
        _check(load().rsp_mcsc_column_sums(self._h, _dp(out)))
        return out

    def column_means(self) -> np.ndarray:
        return self._out(load().rsp_mcsc_column_means, self.ncol)

    def set_gather(self, mode: str) -> None:
        """'d2h' (default) | 'rccl' | 'stores' (rsp_mcsc_set_gather)."""
        _check(load().rsp_mcsc_set_gather(self._h, GATHER_MODES[mode]))

    def set_launch(self, mode: str) -> None:
        """'serial' | 'workers' (rsp_mcsc_set_launch)."""
        _check(load().rsp_mcsc_set_launch(self._h, LAUNCH_MODES[mode]))

    def config(self) -> dict:
        out = np.zeros(4, dtype=np.int32)
        _check(load().rsp_mcsc_config(self._h, _ip(out)))
        inv_g = {v: k for k, v in GATHER_MODES.items()}
        inv_l = {v: k for k, v in LAUNCH_MODES.items()}
        return {"gather": inv_g[int(out[0])], "launch": inv_l[int(out[1])], "workers": int(out[2]), "comms": int(out[3])}

    def result_buffer(self) -> np.ndarray:
        """The handle's page-locked result vector as a numpy view (valid until the next call or close)."""
        ptr = load().rsp_mcsc_result_buffer(self._h)
        return np.ctypeslib.as_array(ptr, shape=(max(self.ncol, 1),))[:self.ncol]

    def last_call_stamps(self) -> dict:
        """Host clock of the last column-sum call in microseconds from its entry (rsp_mcsc_last_call_stamps)."""
        G = self.dims()[2]
        us = np.zeros(1 + 4 * G, dtype=np.float64)
        _check(load().rsp_mcsc_last_call_stamps(self._h, _dp(us), us.size))
        per = us[1:].reshape(G, 4)
        return {"call_us": float(us[0]), "begin_us": per[:, 0].tolist(), "enqueued_us": per[:, 1].tolist(),
                "done_us": per[:, 2].tolist(), "copied_us": per[:, 3].tolist()}

    def shard_kernel_ms(self, k: int, reps: int = 20) -> float:
        ms = ctypes.c_float(0.0)
        _check(load().rsp_mcsc_shard_kernel_ms(self._h, int(k), int(reps), ctypes.byref(ms)))
        return float(ms.value)

    def row_sums(self) -> np.ndarray:
        return self._out(load().rsp_mcsc_row_sums, self.nrow)

    def row_sums_into(self, out: np.ndarray) -> np.ndarray:
        """rsp_mcsc_row_sums into a caller's float64 vector of nrow entries (no allocation per call)."""
        _check(load().rsp_mcsc_row_sums(self._h, _dp(out)))
        return out

    def row_means(self) -> np.ndarray:
        return self._out(load().rsp_mcsc_row_means, self.nrow)

    def dims(self):
        """(nrow, ncol, shards) as the native handle knows them (rsp_mcsc_dims)."""
        a, b, g = ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int32(0)
        _check(load().rsp_mcsc_dims(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(g)))
        return a.value, b.value, g.value

    def shard_info(self, k: int) -> dict:
        """Column range, column-sum form and entries of shard k (rsp_mcsc_shard_info)."""
        out = np.zeros(4, dtype=np.int32)
        _check(load().rsp_mcsc_shard_info(self._h, int(k), _ip(out)))
        return {"c0": int(out[0]), "c1": int(out[1]), "form": DeviceCSC.COLUMN_FORMS[int(out[2])], "nnz": int(out[3])}

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            load().rsp_mcsc_free(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceCSC:
    """Device-resident dgCMatrix (slots x / i / p / Dim): upload once, sum many."""

    def __init__(self, x, p, dim, i=None, device: int = 0):
        x = np.ascontiguousarray(x, dtype=np.float64)
        p = np.ascontiguousarray(p, dtype=np.int32)
        i = None if i is None else np.ascontiguousarray(i, dtype=np.int32)
        self.nrow, self.ncol = int(dim[0]), int(dim[1])
        self.nnz = int(x.size)
        self._h = ctypes.c_void_p()
        _check(load().rsp_csc_upload(_dp(x), _ip(i), _ip(p), self.nrow, self.ncol, self.nnz,
                                     device, ctypes.byref(self._h)))

    def column_sums(self) -> np.ndarray:
        out = np.empty(self.ncol, dtype=np.float64)
        _check(load().rsp_csc_column_sums(self._h, _dp(out)))
        return out

    def column_means(self) -> np.ndarray:
        out = np.empty(self.ncol, dtype=np.float64)
        _check(load().rsp_csc_column_means(self._h, _dp(out)))
        return out

    def crossprod(self) -> np.ndarray:
        """Matrix::crossprod(): dense ncol x ncol t(A) %*% A."""
        out = np.empty((self.ncol, self.ncol), dtype=np.float64, order="F")
        _check(load().rsp_csc_crossprod(self._h, _dp(out)))
        return out

    def row_sums(self) -> np.ndarray:
        out = np.empty(self.nrow, dtype=np.float64)
        _check(load().rsp_csc_row_sums(self._h, _dp(out)))
        return out

    def row_means(self) -> np.ndarray:
        out = np.empty(self.nrow, dtype=np.float64)
        _check(load().rsp_csc_row_means(self._h, _dp(out)))
        return out

    COLUMN_FORMS = ("general kernels", "snapped", "lean", "columns")

    def column_form(self) -> str:
        """Which form this handle's column sums take (rsp_csc_column_form): decided at upload."""
        return self.COLUMN_FORMS[int(load().rsp_csc_column_form(self._h))]

    def set_planned(self, on: bool) -> None:
        """False: the general kernels whatever the upload's plan says (rsp_csc_set_planned; A/B)."""
        _check(load().rsp_csc_set_planned(self._h, int(bool(on))))

    def dims(self):
        nr, nc, nz = ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int64(0)
        _check(load().rsp_csc_dims(self._h, ctypes.byref(nr), ctypes.byref(nc), ctypes.byref(nz)))
        return nr.value, nc.value, nz.value

    ROW_FORMS = ("none", "direct", "partition", "two-level", "segments")

    def row_form(self) -> str:
        """Which form this handle's row sums have taken (rsp_csc_row_form): "none" before the first call."""
        return self.ROW_FORMS[int(load().rsp_csc_row_form(self._h))]

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            load().rsp_csc_free(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# -------------------------------------------------------------- device paths
def _stream_ptr(stream=None):
    import torch
    s = torch.cuda.current_stream() if stream is None else stream
    return ctypes.c_void_p(s.cuda_stream)


def workspace_bytes(ncol: int, nnz: int) -> int:
    return int(load().rsp_column_sums_workspace_bytes(int(ncol), int(nnz)))


def alloc_workspace(ncol: int, nnz: int, device="cuda"):
    import torch
    return torch.empty(workspace_bytes(ncol, nnz), dtype=torch.uint8, device=device)


def column_sums_device(x_t, p_t, out_t=None, workspace=None, stream=None, nrow_for_means=None):
    """columnSums on torch-owned HBM buffers; enqueued on the current torch stream."""
    import torch
    assert x_t.dtype == torch.float64 and p_t.dtype == torch.int32
    assert x_t.is_cuda and p_t.is_cuda and x_t.is_contiguous() and p_t.is_contiguous()
    ncol, nnz = p_t.numel() - 1, x_t.numel()
    if out_t is None:
        out_t = torch.empty(ncol, dtype=torch.float64, device=x_t.device)
    if workspace is None:
        workspace = alloc_workspace(ncol, nnz, x_t.device)
    L = load()
    if nrow_for_means is None:
        _check(L.rsp_column_sums_device(x_t.data_ptr(), p_t.data_ptr(), ncol, nnz, out_t.data_ptr(),
                                        workspace.data_ptr(), workspace.numel(), _stream_ptr(stream)))
    else:
        _check(L.rsp_column_means_device(x_t.data_ptr(), p_t.data_ptr(), int(nrow_for_means), ncol,
                                         nnz, out_t.data_ptr(), workspace.data_ptr(),
                                         workspace.numel(), _stream_ptr(stream)))
    return out_t


def row_sums_device(x_t, i_t, nrow: int, out_t=None, workspace=None, stream=None, ncol_for_means=None):
    """Matrix::rowSums / rowMeans on torch-owned HBM buffers (x, i); p is not needed."""
    import torch
    assert x_t.dtype == torch.float64 and i_t.dtype == torch.int32 and x_t.numel() == i_t.numel()
    nnz = x_t.numel()
    L = load()
    if out_t is None:
        out_t = torch.empty(nrow, dtype=torch.float64, device=x_t.device)
    if workspace is None:
        nbytes = int(L.rsp_row_sums_workspace_bytes(int(nrow), int(nnz)))
        if nbytes == 0:
            _check(RSP_ERR_HIP)
        workspace = torch.empty(nbytes, dtype=torch.uint8, device=x_t.device)
    if ncol_for_means is None:
        _check(L.rsp_row_sums_device(x_t.data_ptr(), i_t.data_ptr(), int(nrow), nnz, out_t.data_ptr(),
                                     workspace.data_ptr(), workspace.numel(), _stream_ptr(stream)))
    else:
        _check(L.rsp_row_means_device(x_t.data_ptr(), i_t.data_ptr(), int(nrow), int(ncol_for_means), nnz,
                                      out_t.data_ptr(), workspace.data_ptr(), workspace.numel(),
                                      _stream_ptr(stream)))
    return out_t


OP_SUM, OP_SUM_SQUARES, OP_SUM_ABS, OP_MAX, OP_MIN, OP_COUNT = 0, 1, 2, 3, 4, 5


def column_reduce_device(x_t, p_t, op: int, out_t=None, workspace=None, stream=None):
    """out[c] = sum over column c of f(x): f = identity / square / abs (rsp_column_reduce_device)."""
    import torch
    ncol, nnz = p_t.numel() - 1, x_t.numel()
    if out_t is None:
        out_t = torch.empty(ncol, dtype=torch.float64, device=x_t.device)
    if workspace is None:
        workspace = alloc_workspace(ncol, nnz, x_t.device)
    _check(load().rsp_column_reduce_device(x_t.data_ptr(), p_t.data_ptr(), ncol, nnz, int(op),
                                           out_t.data_ptr(), workspace.data_ptr(), workspace.numel(),
                                           _stream_ptr(stream)))
    return out_t


def in_rows_workspace_bytes(nrow: int, ncol: int, nnz: int) -> int:
    """Workspace of the row-restricted sums: the column sums' plus the slice form's guard flag."""
    return int(load().rsp_column_sums_in_rows_workspace_bytes(int(nrow), int(ncol), int(nnz)))


IN_ROWS_FORMS = ("L1", "LDS", "L2", "slices")


def in_rows_form(nrow: int, ncol: int, nnz: int, workspace_bytes: int | None = None) -> str:
    """Which form rsp_column_sums_in_rows_device takes at these sizes (rsp_column_sums_in_rows_form)."""
    if workspace_bytes is None:
        workspace_bytes = in_rows_workspace_bytes(nrow, ncol, nnz)
    f = int(load().rsp_column_sums_in_rows_form(int(nrow), int(ncol), int(nnz), int(workspace_bytes)))
    if f < 0:
        raise ValueError("sizes out of range")
    return IN_ROWS_FORMS[f]


def set_row_segments(mode) -> None:
    """Handles' row sums: 0 never the segments form, 1 where it is the faster one (default), 2 wherever it is
    possible (tests) (rsp_debug_set "row_segments").  Looked at when a handle's row sums are asked for the first time."""
    debug_set("row_segments", int(mode))


def set_row_slices(on) -> None:
    """0 / False: row-restricted sums over more than 2^20 rows always probe the bitmap in L2; 1 / True: the slice-major
    form where it is the faster one (default); 2: wherever it is possible at all (tests) (rsp_debug_set "row_slices")."""
    debug_set("row_slices", int(on))


def row_set_bitmap(rows, nrow: int) -> np.ndarray:
    """Sorted row set (as the reference's restricted iterators take it) -> bitmap of nrow bits."""
    bits = np.zeros((int(nrow) + 31) // 32, dtype=np.uint32)
    rows = np.asarray(rows, dtype=np.int64)
    np.bitwise_or.at(bits, rows >> 5, (np.uint32(1) << (rows & 31).astype(np.uint32)))
    return bits


def column_sums_in_rows_device(x_t, i_t, p_t, nrow: int, bitmap_t, complement: bool = False, out_t=None,
                               workspace=None, stream=None):
    """Column sums restricted to entries whose row is (not) in the set (rsp_column_sums_in_rows_device)."""
    import torch
    ncol, nnz = p_t.numel() - 1, x_t.numel()
    if out_t is None:
        out_t = torch.empty(ncol, dtype=torch.float64, device=x_t.device)
    if workspace is None:
        workspace = torch.empty(in_rows_workspace_bytes(nrow, ncol, nnz), dtype=torch.uint8, device=x_t.device)
    _check(load().rsp_column_sums_in_rows_device(x_t.data_ptr(), i_t.data_ptr(), p_t.data_ptr(), int(nrow),
                                                 ncol, nnz, bitmap_t.data_ptr(), int(bool(complement)),
                                                 out_t.data_ptr(), workspace.data_ptr(), workspace.numel(),
                                                 _stream_ptr(stream)))
    return out_t


def crossprod_form(nrow: int, ncol: int, nnz: int) -> str:
    """Which form rsp_crossprod_device (with a workspace) takes at these sizes: "tall" (matrix cores) or "exact"."""
    f = int(load().rsp_crossprod_form(int(nrow), int(ncol), int(nnz)))
    if f < 0:
        raise RspError(RSP_ERR_HIP, load().rsp_last_error().decode())
    return ("exact", "tall")[f]


def crossprod_device(x_t, i_t, p_t, nrow, out_t=None, workspace=None, stream=None, tiles=False):
    """Dense t(A) %*% A on torch-owned buffers; returns an ncol x ncol tensor (symmetric).
    ``tiles=True`` runs the scratch-free tile kernel (no workspace); otherwise the row-major
    path, with a workspace of rsp_crossprod_workspace_bytes allocated here unless given."""
    import torch
    L = load()
    ncol, nnz = p_t.numel() - 1, x_t.numel()
    if out_t is None:
        out_t = torch.empty((ncol, ncol), dtype=torch.float64, device=x_t.device)
    ws_ptr, ws_bytes = None, 0
    if not tiles:
        if workspace is None:
            nbytes = int(L.rsp_crossprod_workspace_bytes(int(nrow), int(ncol), int(nnz)))
            if nbytes == 0:
                raise RspError(RSP_ERR_HIP, L.rsp_last_error().decode())
            workspace = torch.empty(nbytes, dtype=torch.uint8, device=x_t.device)
        ws_ptr, ws_bytes = workspace.data_ptr(), workspace.numel()
    _check(L.rsp_crossprod_device(x_t.data_ptr(), i_t.data_ptr(), p_t.data_ptr(), int(nrow), ncol, nnz,
                                  out_t.data_ptr(), ws_ptr, ws_bytes, _stream_ptr(stream)))
    return out_t


def prepared_column_sums(x_t, p_t, out_t, workspace, stream=None):
    """Pre-bound launcher of rsp_column_sums_device for hot loops (bench): the ctypes
    arguments are converted once, each call is a single foreign call."""
    L = load()
    ncol, nnz = p_t.numel() - 1, x_t.numel()
    fn = L.rsp_column_sums_device
    args = (ctypes.c_void_p(x_t.data_ptr()), ctypes.c_void_p(p_t.data_ptr()), ctypes.c_int32(ncol),
            ctypes.c_int64(nnz), ctypes.c_void_p(out_t.data_ptr()), ctypes.c_void_p(workspace.data_ptr()),
            ctypes.c_size_t(workspace.numel()), _stream_ptr(stream))

    def launch():
        rc = fn(*args)
        if rc != RSP_OK:
            _check(rc)
    return launch


class ColumnSumsPlan:
    """Inspector-executor plan of a column-sum call (rsp_column_sums_plan_*): made once from p[], then every call is
    one launch without column search, carries or fix-up -- where the matrix allows it (``snapped``); otherwise the
    executor runs the general kernels and needs their workspace.
    p a host array: inspected on the host (rsp_column_sums_plan_create).  p a device tensor: inspected ON THE DEVICE, on
    ``stream``, without any synchronisation (rsp_column_sums_plan_create_device); until the host has seen the result
    (``ready()``; ``wait()`` blocks) calls run the general kernels, and the attributes ``form`` / ``lean`` / ``snapped``
    / ``columns`` / ``inspect_ms`` ... wait for it when they are read."""

    _INFO = ("form", "lean", "columns", "snapped", "nchunks", "chunk_elems", "max_skip", "inspect_ms")

    def __init__(self, p, nnz: int = None, device: int = 0, stream=None):
        L = load()
        self._h = ctypes.c_void_p()
        self._info = None
        if isinstance(p, np.ndarray):
            p = np.ascontiguousarray(p, dtype=np.int32)
            self.ncol = len(p) - 1
            self.nnz = int(p[-1]) if nnz is None else int(nnz)
            self.device_made = False
            _check(L.rsp_column_sums_plan_create(_ip(p), self.ncol, self.nnz, int(device), ctypes.byref(self._h)))
            self._load_info()
        else:   # a device tensor of offsets
            self.ncol = p.numel() - 1
            if nnz is None:
                raise ValueError("nnz is needed with device offsets")
            self.nnz = int(nnz)
            self.device_made = True
            _check(L.rsp_column_sums_plan_create_device(p.data_ptr(), self.ncol, self.nnz, _stream_ptr(stream),
                                                        ctypes.byref(self._h)))

    def _load_info(self):
        info = np.zeros(4, dtype=np.int32)
        ms = ctypes.c_double(0)
        _check(load().rsp_column_sums_plan_info(self._h, _ip(info), ctypes.byref(ms)))   # (waits for a device-made plan)
        # form 3 = columns (all columns long: one workgroup per column), 2 = lean (all columns short: one launch,
        # reference bits for every column), 1 = snapped (one launch), 0 = the general kernels behind the same entry
        form = int(info[0])
        self._info = {"form": form, "lean": form == 2, "columns": form == 3, "snapped": form >= 1,
                      "nchunks": int(info[1]), "chunk_elems": int(info[2]), "max_skip": int(info[3]),
                      "inspect_ms": float(ms.value)}

    def __getattr__(self, name):
        if name in ColumnSumsPlan._INFO:
            if self.__dict__.get("_info") is None:
                self._load_info()
            return self.__dict__["_info"][name]
        raise AttributeError(name)

    def ready(self) -> bool:
        """Has the host seen the inspection's result (always True for a host-made plan)?  Never blocks."""
        return int(load().rsp_column_sums_plan_ready(self._h)) == 1

    def wait(self) -> "ColumnSumsPlan":
        _check(load().rsp_column_sums_plan_wait(self._h))
        return self

    def image(self, what: int) -> np.ndarray:
        """The plan's image as the device holds it (rsp_debug_plan_image): what = 0 the snapped records, 1 the lean
        headers + 16-bit offsets; uint32 words, empty when the plan has no such image."""
        n = ctypes.c_size_t(0)
        _check(load().rsp_debug_plan_image(self._h, int(what), None, 0, ctypes.byref(n)))
        out = np.zeros(n.value // 4, dtype=np.uint32)
        if n.value:
            _check(load().rsp_debug_plan_image(self._h, int(what), out.ctypes.data, out.nbytes, ctypes.byref(n)))
        return out

    def column_sums(self, x_t, p_t, out_t=None, workspace=None, stream=None, nrow_for_means: int = 0):
        import torch
        assert x_t.numel() == self.nnz and p_t.numel() == self.ncol + 1
        if out_t is None:
            out_t = torch.empty(self.ncol, dtype=torch.float64, device=x_t.device)
        # (a device-made plan whose result has not been looked at yet may still answer with the general kernels)
        info = self.__dict__.get("_info")
        if workspace is None and (info is None or not info["snapped"]):
            workspace = alloc_workspace(self.ncol, self.nnz, x_t.device)
        _check(load().rsp_column_sums_planned_device(
            self._h, x_t.data_ptr(), p_t.data_ptr(), self.ncol, self.nnz, int(nrow_for_means), out_t.data_ptr(),
            workspace.data_ptr() if workspace is not None else None,
            workspace.numel() if workspace is not None else 0, _stream_ptr(stream)))
        return out_t

    def prepared(self, x_t, p_t, out_t, workspace=None, stream=None):
        """Pre-bound launcher for hot loops (bench): one foreign call per launch."""
        fn = load().rsp_column_sums_planned_device
        args = (self._h, ctypes.c_void_p(x_t.data_ptr()), ctypes.c_void_p(p_t.data_ptr()),
                ctypes.c_int32(p_t.numel() - 1), ctypes.c_int64(x_t.numel()), ctypes.c_int32(0),
                ctypes.c_void_p(out_t.data_ptr()),
                ctypes.c_void_p(workspace.data_ptr() if workspace is not None else None),
                ctypes.c_size_t(workspace.numel() if workspace is not None else 0), _stream_ptr(stream))

        def launch():
            rc = fn(*args)
            if rc != RSP_OK:
                _check(rc)
        return launch

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            load().rsp_column_sums_plan_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def column_sums_device_timed(x_t, p_t, out_t, workspace, reps: int, stream=None) -> float:
    """Mean ms per call, hipEvents recorded on the launch stream inside the library."""
    ms = ctypes.c_float(0)
    ncol, nnz = p_t.numel() - 1, x_t.numel()
    _check(load().rsp_column_sums_device_timed(x_t.data_ptr(), p_t.data_ptr(), ncol, nnz,
                                               out_t.data_ptr(), workspace.data_ptr(),
                                               workspace.numel(), _stream_ptr(stream), int(reps),
                                               ctypes.byref(ms)))
    return float(ms.value)


def exclusive_scan_device(in_t, out_t=None, stream=None):
    """Exclusive prefix sum of an int32 tensor through the library's hand-written scan (rsp_debug_exclusive_scan_device)."""
    import torch
    assert in_t.dtype == torch.int32 and in_t.is_contiguous()
    if out_t is None:
        out_t = torch.empty_like(in_t)
    _check(load().rsp_debug_exclusive_scan_device(in_t.data_ptr(), out_t.data_ptr(), in_t.numel(), _stream_ptr(stream)))
    return out_t


def read_ceiling_device(x_t, reps: int = 5, stream=None) -> float:
    """Mean ms per launch of the read-only kernel with the column sums' access shape over x_t
    (rsp_debug_read_ceiling_device): the practical ceiling of this device for this stream."""
    import torch
    sink = torch.zeros(1, dtype=torch.float64, device=x_t.device)
    ms = ctypes.c_float(0)
    _check(load().rsp_debug_read_ceiling_device(x_t.data_ptr(), x_t.numel(), sink.data_ptr(), _stream_ptr(stream),
                                                int(reps), ctypes.byref(ms)))
    return float(ms.value)


def gen_values_device(x_t, seed: int, first_idx: int = 0, kind: int = 0, stream=None):
    _check(load().rsp_gen_values_device(x_t.data_ptr(), x_t.numel(), int(seed), int(first_idx),
                                        int(kind), _stream_ptr(stream)))
    return x_t


def gen_row_indices_device(i_t, p_t, nrow: int, seed: int, stream=None):
    _check(load().rsp_gen_row_indices_device(i_t.data_ptr(), p_t.data_ptr(), int(nrow), p_t.numel() - 1,
                                             int(seed), _stream_ptr(stream)))
    return i_t


# ------------------------------------------------------------- partitioning
def partition_columns(p, nparts: int) -> np.ndarray:
    p = np.ascontiguousarray(p, dtype=np.int32)
    bounds = np.empty(nparts + 1, dtype=np.int32)
    _check(load().rsp_partition_columns(_ip(p), len(p) - 1, int(nparts), _ip(bounds)))
    return bounds


def rebase_offsets(p, c0: int, c1: int) -> np.ndarray:
    p = np.ascontiguousarray(p, dtype=np.int32)
    out = np.empty(c1 - c0 + 1, dtype=np.int32)
    _check(load().rsp_rebase_offsets(_ip(p), int(c0), int(c1), _ip(out)))
    return out


# --------------------------------------------------------------------- RCCL
def rccl_info() -> dict:
    """{"version": ncclGetVersion of the RCCL this process runs, "library": the file it is mapped from} (rsp_rccl_info)."""
    v = ctypes.c_int(0)
    buf = ctypes.create_string_buffer(512)
    _check(load().rsp_rccl_info(ctypes.byref(v), buf, 512))
    return {"version": int(v.value), "library": buf.value.decode("utf-8", "replace")}


def comm_unique_id() -> bytes:
    buf = ctypes.create_string_buffer(UNIQUE_ID_BYTES)
    _check(load().rsp_comm_unique_id(buf))
    return buf.raw


class Comm:
    """One RCCL communicator per process (one process per GPU)."""

    def __init__(self, unique_id: bytes, nranks: int, rank: int, device: int):
        assert len(unique_id) == UNIQUE_ID_BYTES
        self.nranks, self.rank = nranks, rank
        self._h = ctypes.c_void_p()
        buf = ctypes.create_string_buffer(unique_id, UNIQUE_ID_BYTES)
        _check(load().rsp_comm_init(buf, nranks, rank, device, ctypes.byref(self._h)))

    def gatherv(self, send_t, recv_t, counts, displs, root: int = 0, stream=None) -> None:
        n = self.nranks
        c_arr = (ctypes.c_int64 * n)(*[int(v) for v in counts])
        d_arr = (ctypes.c_int64 * n)(*[int(v) for v in displs])
        _check(load().rsp_comm_gatherv(self._h, send_t.data_ptr(), send_t.numel(),
                                       recv_t.data_ptr() if recv_t is not None else None,
                                       c_arr, d_arr, root, _stream_ptr(stream)))

    def prepared_gatherv(self, send_t, recv_t, counts, displs, root: int = 0, stream=None):
        """Pre-bound gatherv for hot loops: argument conversion happens once."""
        n = self.nranks
        c_arr = (ctypes.c_int64 * n)(*[int(v) for v in counts])
        d_arr = (ctypes.c_int64 * n)(*[int(v) for v in displs])
        fn = load().rsp_comm_gatherv
        args = (self._h, ctypes.c_void_p(send_t.data_ptr()), ctypes.c_int64(send_t.numel()),
                ctypes.c_void_p(recv_t.data_ptr() if recv_t is not None else None), c_arr, d_arr,
                ctypes.c_int(root), _stream_ptr(stream))

        def run():
            rc = fn(*args)
            if rc != RSP_OK:
                _check(rc)
        return run

    def reduce_rows(self, partial_t, result_t, root: int = 0, workspace=None, ncol_for_means: int = 0,
                    stream=None):
        """Partial row sums of every rank (nrow doubles each) -> their sum in rank order on `root`
        (rsp_comm_reduce_rows).  result_t is only needed on the root."""
        import torch
        nrow = partial_t.numel()
        if workspace is None:
            workspace = torch.empty(reduce_rows_workspace_bytes(self.nranks, nrow), dtype=torch.uint8,
                                    device=partial_t.device)
        _check(load().rsp_comm_reduce_rows(self._h, partial_t.data_ptr(), nrow, int(ncol_for_means),
                                           result_t.data_ptr() if result_t is not None else None,
                                           workspace.data_ptr(), workspace.numel(), root, _stream_ptr(stream)))
        return result_t

    def close(self) -> None:
        if self._h:
            load().rsp_comm_destroy(self._h)
            self._h = ctypes.c_void_p()


IPC_HANDLE_BYTES = 64


class SharedResult:
    """The root's result buffer of the direct-write gather (rsp_shared_result_*): rank `root` allocates it and
    exports a 64-byte handle, the other rank processes map it; ``tensor()`` is a float64 view of n doubles
    (torch only wraps the pointer: the memory belongs to this object)."""

    def __init__(self, n: int, handle: bytes = None):
        import torch   # noqa: F401
        self.n, self.owner = int(n), handle is None
        self._p = ctypes.c_void_p()
        if self.owner:
            buf = ctypes.create_string_buffer(IPC_HANDLE_BYTES)
            _check(load().rsp_shared_result_alloc(self.n * 8, ctypes.byref(self._p), buf))
            self.handle = buf.raw
        else:
            assert len(handle) == IPC_HANDLE_BYTES
            self.handle = bytes(handle)
            buf = ctypes.create_string_buffer(self.handle, IPC_HANDLE_BYTES)
            _check(load().rsp_shared_result_open(buf, ctypes.byref(self._p)))

    @property
    def ptr(self) -> int:
        return int(self._p.value)

    def read(self, stream=None) -> np.ndarray:
        """The n doubles as a host array (rsp_shared_result_read; waits for `stream`)."""
        out = np.empty(self.n, dtype=np.float64)
        _check(load().rsp_shared_result_read(self._p, 0, out.ctypes.data, out.nbytes, _stream_ptr(stream)))
        return out

    def close(self) -> None:
        if self._p:
            load().rsp_shared_result_close(self._p, int(self.owner))
            self._p = ctypes.c_void_p()


class SharedHostVector:
    """n doubles in POSIX shared memory, mapped and page-locked in this process (rsp_shared_host_open): the rank processes'
    copy engines write their slices into it, the root reads the whole (`array`)."""

    def __init__(self, name: str, n: int, create: bool):
        self.name, self.n, self.bytes, self.create = name, int(n), max(8, 8 * int(n)), bool(create)
        p = ctypes.c_void_p()
        _check(load().rsp_shared_host_open(name.encode(), self.bytes, int(self.create), ctypes.byref(p)))
        self.ptr = int(p.value)
        self.array = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_double)), shape=(max(self.n, 1),))[:self.n]

    def copy_from_device(self, src_t, offset: int, stream=None) -> None:
        """enqueue: src_t (float64, device) -> vector[offset : offset + len] on `stream`"""
        _check(load().rsp_copy_to_host_async(src_t.data_ptr(), ctypes.c_void_p(self.ptr + 8 * int(offset)), int(src_t.numel()),
                                             _stream_ptr(stream)))

    def close(self) -> None:
        if getattr(self, "ptr", 0):
            self.array = None
            load().rsp_shared_host_close(ctypes.c_void_p(self.ptr), self.bytes, self.name.encode() if self.create else None)
            self.ptr = 0


def device_can_access_peer(device: int, peer: int) -> bool:
    can = ctypes.c_int(0)
    _check(load().rsp_device_can_access_peer(int(device), int(peer), ctypes.byref(can)))
    return bool(can.value)


class HostBarrier:
    """Barrier between the rank processes of one node through POSIX shared memory (rsp_host_barrier_*)."""

    def __init__(self, name: str, nranks: int, rank: int):
        self._h = ctypes.c_void_p()
        _check(load().rsp_host_barrier_create(name.encode(), int(nranks), int(rank), ctypes.byref(self._h)))
        self._wait = load().rsp_host_barrier_wait

    def wait(self, timeout: float = 60.0) -> None:
        rc = self._wait(self._h, timeout)
        if rc != RSP_OK:
            _check(rc)

    def close(self) -> None:
        if self._h:
            load().rsp_host_barrier_destroy(self._h)
            self._h = ctypes.c_void_p()


def reduce_rows_workspace_bytes(nranks: int, nrow: int) -> int:
    return int(load().rsp_comm_reduce_rows_workspace_bytes(int(nranks), int(nrow)))


def add_partials_device(parts_t, out_t=None, ncol_for_means: int = 0, stream=None):
    """parts_t: (nparts, n) contiguous tensor of partial sums -> their sum over the parts in part order
    (rsp_add_partials_device): what rsp_comm_reduce_rows computes, for one process holding all shards."""
    import torch
    assert parts_t.dim() == 2 and parts_t.is_contiguous() and parts_t.dtype == torch.float64
    nparts, n = parts_t.shape
    if out_t is None:
        out_t = torch.empty(n, dtype=torch.float64, device=parts_t.device)
    _check(load().rsp_add_partials_device(parts_t.data_ptr(), nparts, n, n, int(ncol_for_means),
                                          out_t.data_ptr(), _stream_ptr(stream)))
    return out_t
