"""Compiles THIS repository's Rcpp layer -- the drop-in header `rcppsparse_amd/host/RcppSparse.h`,
the exported `columnSums` (`rpkg/src/columnSums.cpp`) and the registration glue
(`rpkg/src/rcpp_glue.cpp`) -- against an API-shaped mock of Rcpp (`tests/mock_rcpp/`, test
scaffolding; neither R nor Rcpp exists in this image) and drives it the way R does: package
init, routine table, `.Call` with a dgCMatrix-like S4 object.  What the boundary must keep
(reference src/RcppExports.cpp:16-34, inst/include/RcppSparse.h:33-42, :398-423): the two C
symbols, arity-1 registration with dynamic lookup off, zero-copy S4 -> Matrix, the missing-slot
message, C++ exceptions surfacing as R errors, and no CPU fallback."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    from rcppsparse_amd import capi
    capi.load()                                   # make sure librcppsparse_hip.so exists
    rpkg = os.path.join(ROOT, "rcppsparse_amd", "host", "rpkg")
    subprocess.run(["bash", os.path.join(rpkg, "assemble.sh")], check=True, stdout=subprocess.DEVNULL)
    exe = str(tmp_path_factory.mktemp("mock") / "driver")
    libdir = os.path.join(ROOT, "rcppsparse_amd")
    mock = os.path.join(ROOT, "tests", "mock_rcpp")
    subprocess.run(["g++", "-std=c++14", "-O1", "-Wall", "-I", mock, "-o", exe,
                    os.path.join(mock, "driver.cpp"), os.path.join(rpkg, "src", "columnSums.cpp"),
                    os.path.join(rpkg, "src", "gpuMatrix.cpp"),
                    os.path.join(rpkg, "src", "rcpp_glue.cpp"), "-L", libdir, "-lrcppsparse_hip",
                    f"-Wl,-rpath,{libdir}"], check=True)
    return exe


def run(exe, mode):
    return subprocess.run([exe, mode], capture_output=True, text=True, timeout=120)


def test_package_registers_the_reference_routine_first(driver):
    """Entry 0 is the reference's one routine (src/RcppExports.cpp:26-29), name and arity; the
    gpuMatrix handle routines follow it."""
    r = run(driver, "registered")
    assert r.returncode == 0
    rows = [ln.split() for ln in r.stdout.splitlines()]
    assert rows[0] == ["_RcppSparse_columnSums", "1"]
    assert rows[1:] == [["_RcppSparse_gpuMatrix", "2"], ["_RcppSparse_gpuColumnSums", "1"],
                        ["_RcppSparse_gpuFree", "1"], ["_RcppSparse_gpuReduce", "2"],
                        ["_RcppSparse_gpuCrossprod", "1"], ["_RcppSparse_gpuMatrixMulti", "2"],
                        ["_RcppSparse_gpuMultiReduce", "2"], ["_RcppSparse_gpuFreeMulti", "1"]]


def test_glue_also_builds_with_global_rostream(tmp_path):
    """reference src/RcppExports.cpp:9-12: a build with -DRCPP_USE_GLOBAL_ROSTREAM has to define
    Rcpp::Rcout / Rcpp::Rcerr in the package; the glue carries those definitions (they are
    compiled here against stand-in declarations, since real Rcpp is not in this image)."""
    rpkg = os.path.join(ROOT, "rcppsparse_amd", "host", "rpkg")
    subprocess.run(["bash", os.path.join(rpkg, "assemble.sh")], check=True, stdout=subprocess.DEVNULL)
    shim = tmp_path / "rostream_decl.h"
    shim.write_text("namespace Rcpp { template <bool B> struct Rostream {}; "
                    "extern Rostream<true>& Rcout; extern Rostream<false>& Rcerr; "
                    "inline Rostream<true>& Rcpp_cout_get() { static Rostream<true> s; return s; } "
                    "inline Rostream<false>& Rcpp_cerr_get() { static Rostream<false> s; return s; } }\n")
    obj = tmp_path / "glue.o"
    subprocess.run(["g++", "-std=c++14", "-c", "-DRCPP_USE_GLOBAL_ROSTREAM", "-include", str(shim),
                    "-I", os.path.join(ROOT, "tests", "mock_rcpp"), "-o", str(obj),
                    os.path.join(rpkg, "src", "rcpp_glue.cpp")], check=True)
    syms = subprocess.run(["nm", "-C", str(obj)], capture_output=True, text=True, check=True).stdout
    assert "Rcpp::Rcout" in syms and "Rcpp::Rcerr" in syms


def test_missing_slot_becomes_an_r_error_with_the_reference_message(driver):
    r = run(driver, "missing_slot")
    assert r.returncode == 0
    assert r.stdout.strip() == "Cannot construct RcppSparse::Matrix from this S4 object"


def test_without_a_gpu_columnSums_is_an_r_error_not_a_cpu_answer(driver):
    from rcppsparse_amd import capi
    if capi.device_count() > 0:
        pytest.skip("a GPU is present")
    r = run(driver, "kat")
    assert r.returncode == 16 and "no HIP device" in r.stdout


def test_without_a_gpu_gpuMatrix_is_an_r_error(driver):
    from rcppsparse_amd import capi
    if capi.device_count() > 0:
        pytest.skip("a GPU is present")
    r = run(driver, "handle_nogpu")
    assert r.returncode == 0 and "no HIP device" in r.stdout


@pytest.mark.gpu
def test_gpu_matrix_external_pointer_handle(driver):
    """gpuMatrix(A) -> external pointer (finalizer = rsp_csc_free); columnSums on it runs on the
    resident copy, does not see later in-place changes of A, survives neither gpuFree nor the
    collector, and a released handle is an R error."""
    r = run(driver, "handle")
    assert r.returncode == 0, (r.returncode, r.stdout + r.stderr)
    assert "gpuMatrix handle ok" in r.stdout


@pytest.mark.gpu
def test_gpu_matrix_handle_methods_and_multi_gpu_handle(driver):
    """The rest of the handle at the R level: colMeans / rowSums / rowMeans /
    crossprod on a "gpuMatrix" (reference RcppSparse.h:131-194 on the resident copy) and the same matrix spread
    over three shards by gpuMatrix(A, devices = c(0, 0, 0)) (rsp_mcsc_*), driven through the registered .Call
    routines: the vignette's 5 x 5 matrix, expected values bit for bit, released handles are R errors, the
    collector's finalizer frees every shard."""
    r = run(driver, "handle_methods")
    assert r.returncode == 0, (r.returncode, r.stdout + r.stderr)
    assert "gpuMatrix methods ok" in r.stdout


@pytest.mark.gpu
def test_dot_call_columnSums_on_the_gpu(driver):
    r = run(driver, "kat")
    assert r.returncode == 0, r.stdout + r.stderr
    assert "columnSums via .Call ok" in r.stdout
