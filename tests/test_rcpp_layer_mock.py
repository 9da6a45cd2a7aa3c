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
                    os.path.join(rpkg, "src", "rcpp_glue.cpp"), "-L", libdir, "-lrcppsparse_hip",
                    f"-Wl,-rpath,{libdir}"], check=True)
    return exe


def run(exe, mode):
    return subprocess.run([exe, mode], capture_output=True, text=True, timeout=120)


def test_package_registers_exactly_the_reference_routine(driver):
    r = run(driver, "registered")
    assert r.returncode == 0 and r.stdout.split() == ["_RcppSparse_columnSums", "1"]


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


@pytest.mark.gpu
def test_dot_call_columnSums_on_the_gpu(driver):
    r = run(driver, "kat")
    assert r.returncode == 0, r.stdout + r.stderr
    assert "columnSums via .Call ok" in r.stdout
