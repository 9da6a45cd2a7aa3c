"""Build helper: compiles librcppsparse_hip.so for gfx950 with hipcc, in-tree."""
from __future__ import annotations

import contextlib
import fcntl
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(_HERE, "librcppsparse_hip.so")
HOST_SEAM_PATH = os.path.join(_HERE, "librcppsparse_host.so")
_SOURCES = ["colsums_kernels.hip", "colsums_rowslices.hip", "inspect_device.hip", "scan.hip", "colsums_kernels.h", "inspect.hpp", "shard_workers.hpp", "rowsums.hip", "crossprod.hip", "capi.hip", "multigpu.cpp", "Makefile",
            os.path.join("..", "..", "include", "rcppsparse_hip.h")]


def _stale(target: str, sources) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(s) and os.path.getmtime(s) > t for s in sources)


@contextlib.contextmanager
def _build_lock(name: str):
    """Serialises builds across processes (8 bench ranks importing at once must not run
    `make` into the same .so concurrently); the waiters re-check staleness afterwards."""
    path = os.path.join(_HERE, f".{name}.lock")
    try:
        fd = os.open(path, os.O_CREAT | os.O_RDWR, 0o644)
    except OSError:          # read-only tree: nothing can be built anyway
        yield
        return
    try:
        fcntl.flock(fd, fcntl.LOCK_EX)
        yield
    finally:
        fcntl.flock(fd, fcntl.LOCK_UN)
        os.close(fd)


def have_hipcc() -> bool:
    return shutil.which("hipcc") is not None or os.path.exists("/opt/rocm/bin/hipcc")


def build_library(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 -shared ... -> rcppsparse_amd/librcppsparse_hip.so"""
    srcs = [os.path.join(CSRC, s) for s in _SOURCES]
    if force or _stale(LIB_PATH, srcs):
        if not have_hipcc():
            raise RuntimeError("hipcc not found: cannot build librcppsparse_hip.so")
        with _build_lock("build_hip"):
            if force or _stale(LIB_PATH, srcs):      # another process may have built it meanwhile
                cmd = ["make", "-j4", "-C", CSRC] + (["-B"] if force else [])
                subprocess.run(cmd, check=True, stdout=None if verbose else subprocess.DEVNULL)
    return LIB_PATH


def build_host_seam(force: bool = False, verbose: bool = False) -> str:
    """g++ build of the Rcpp-free host mirror (RcppSparse::Matrix over raw views)."""
    host = os.path.join(_HERE, "host")
    srcs = ([os.path.join(host, f) for f in os.listdir(host) if os.path.isfile(os.path.join(host, f))]
            if os.path.isdir(host) else [])
    if not srcs:
        raise RuntimeError("host/ sources missing")
    # (the seam only links the C-ABI library by name: a newer librcppsparse_hip.so does not make it
    # stale, and the host Makefile does not depend on it either)
    srcs.append(os.path.join(_HERE, "..", "include", "rcppsparse_hip.h"))
    if force or _stale(HOST_SEAM_PATH, srcs):
        with _build_lock("build_host"):
            if force or _stale(HOST_SEAM_PATH, srcs):
                cmd = ["make", "-C", host] + (["-B"] if force else [])
                subprocess.run(cmd, check=True, stdout=None if verbose else subprocess.DEVNULL)
    return HOST_SEAM_PATH
