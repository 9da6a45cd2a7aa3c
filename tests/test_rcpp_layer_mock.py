"""Compiles THIS repository's Rcpp layer -- the drop-in header `rcppsparse_amd/host/RcppSparse.h`,
the exported `columnSums` (`rpkg/src/columnSums.cpp`) and the registration glue
(`rpkg/src/rcpp_glue.cpp`) -- against an API-shaped mock of Rcpp (`tests/mock_rcpp/`, test
scaffolding; neither R nor Rcpp exists in this image) and drives it the way R does: package
init, routine table, `.Call` with a dgCMatrix-like S4 object.  What the boundary must keep
(reference src/RcppExports.cpp:16-34, inst/include/RcppSparse.h:33-42, :398-423): the two C
symbols, arity-1 registration with dynamic lookup off, zero-copy S4 -> Matrix, the missing-slot
message, C++ exceptions surfacing as R errors; on a machine without any GPU the host loop answers (above the
C ABI, which itself never falls back) unless a GPU is required."""
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


def run(exe, mode, require_gpu="1"):
    env = dict(os.environ)
    env.pop("RCPPSPARSE_REQUIRE_GPU", None)
    if require_gpu is not None:
        env["RCPPSPARSE_REQUIRE_GPU"] = require_gpu
    return subprocess.run([exe, mode], capture_output=True, text=True, timeout=120, env=env)


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
                        ["_RcppSparse_gpuMultiReduce", "2"], ["_RcppSparse_gpuFreeMulti", "1"],
                        ["_RcppSparse_columnSumsBackend", "1"]]


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


def test_without_a_gpu_and_a_gpu_required_columnSums_is_an_r_error(driver):
    """RCPPSPARSE_REQUIRE_GPU=1 (what this suite, bench.py and smoke() run under): no CPU answer."""
    from rcppsparse_amd import capi
    if capi.device_count() > 0:
        pytest.skip("a GPU is present")
    r = run(driver, "kat", require_gpu="1")
    assert r.returncode == 16 and "no HIP device" in r.stdout
    r = run(driver, "backend", require_gpu="1")
    assert r.stdout.split() == ["none", "none"]


KAT_HEX = ["0x0p+0", "0x1.a3d70a3d70a3dp-2", "0x1.6666666666666p-2", "0x1.35c28f5c28f5cp+0", "0x1.0a3d70a3d70a4p-2"]


def test_without_a_gpu_columnSums_answers_on_the_host_like_the_reference(driver):
    """SURVEY.md 8b "CPU fallback selected when no device": the reference's columnSums always answers
    (src/example.cpp:26-32).  Through .Call on a machine without a GPU: the five sums of the vignette's matrix
    (Documentation.Rmd:213-216) bit for bit, a plain vector without attributes, columnSumsBackend() = "cpu";
    options(RcppSparse.require_gpu = TRUE) turns the same call into the R error, FALSE overrides the environment."""
    from rcppsparse_amd import capi
    if capi.device_count() > 0:
        pytest.skip("a GPU is present: the host loop is never selected")
    r = run(driver, "kat_cpu", require_gpu=None)
    assert r.returncode == 0, (r.returncode, r.stdout + r.stderr)
    lines = r.stdout.splitlines()
    assert lines[0].split() == KAT_HEX
    assert lines[1] == "backend now=cpu last=cpu"
    assert lines[2].startswith("required: RcppSparse columnSums (HIP): no HIP device") and lines[2].endswith("now=none")
    for off in ("0", ""):
        assert run(driver, "kat_cpu", require_gpu=off).returncode == 0      # "0" and empty do not require a GPU
    r = run(driver, "kat_cpu", require_gpu="1")                             # the environment requires one ...
    assert r.returncode == 72 and "no HIP device" in r.stdout               # ... and the first call is the R error


def test_offload_threshold_settings_without_a_gpu():
    """SURVEY.md section 5 "min-nnz threshold for GPU offload": the setting's three sources in their order --
    options(RcppSparse.min_nnz = n), RCPPSPARSE_MIN_NNZ, the measured default -- and the decision itself.  Without a GPU
    the host loop answers whatever the threshold says, and a required GPU is "none" (an R error) whatever it says."""
    from rcppsparse_amd import capi, hostseam
    keep = os.environ.pop("RCPPSPARSE_MIN_NNZ", None)
    try:
        assert hostseam.min_nnz() == 250_000 and hostseam.min_nnz(17) == 17 and hostseam.min_nnz(0) == 0
        os.environ["RCPPSPARSE_MIN_NNZ"] = "4000"
        assert hostseam.min_nnz() == 4000 and hostseam.min_nnz(17) == 17          # the R option wins over the environment
        os.environ["RCPPSPARSE_MIN_NNZ"] = "-3"
        assert hostseam.min_nnz() == 0
    finally:
        os.environ.pop("RCPPSPARSE_MIN_NNZ", None)
        if keep is not None:
            os.environ["RCPPSPARSE_MIN_NNZ"] = keep
    if capi.device_count() == 0:
        for nnz in (0, 10, 10**9):
            assert hostseam.backend_for(nnz, require_gpu=0, min_nnz=1000) == "cpu"
            assert hostseam.backend_for(nnz, require_gpu=1, min_nnz=1000) == "none"
    else:
        assert hostseam.backend_for(999, require_gpu=0, min_nnz=1000) == "cpu"       # below the threshold: the host loop
        assert hostseam.backend_for(1000, require_gpu=0, min_nnz=1000) == "hip"
        assert hostseam.backend_for(10, require_gpu=0, min_nnz=0) == "hip"           # 0: every matrix goes to the device
        assert hostseam.backend_for(10, require_gpu=1, min_nnz=1000) == "hip"        # a required GPU overrides the threshold


@pytest.mark.gpu
def test_offload_threshold_through_dot_call_on_a_gpu_box(driver):
    """The exported columnSums(A) on a machine WITH a GPU, through .Call: the vignette's matrix (5 stored entries) is answered
    by the host loop under the default threshold -- the drop-in is never slower than the reference on the reference's own
    examples (README.md:33-38) --, by the device with options(RcppSparse.min_nnz = 0) or a threshold of 5, by the host loop
    with 6; options(RcppSparse.require_gpu = TRUE) sends it to the device whatever the threshold; RCPPSPARSE_MIN_NNZ=0 in the
    environment does what the option does; and RCPPSPARSE_REQUIRE_GPU=1 (this suite's setting) keeps every call on the
    device.  The sums are the reference's bits on both paths."""
    r = run(driver, "min_nnz", require_gpu=None)
    assert r.returncode == 0, (r.returncode, r.stdout + r.stderr)
    assert [ln.split() for ln in r.stdout.splitlines()] == [["default", "cpu"], ["min_nnz=0", "hip"], ["min_nnz=6", "cpu"],
                                                            ["min_nnz=5", "hip"], ["min_nnz=6+require_gpu", "hip"]]
    env = dict(os.environ, RCPPSPARSE_MIN_NNZ="0")
    env.pop("RCPPSPARSE_REQUIRE_GPU", None)
    r = subprocess.run([driver, "min_nnz"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0 and r.stdout.splitlines()[0].split() == ["default", "hip"]
    r = run(driver, "min_nnz", require_gpu="1")
    assert r.returncode == 0 and all(ln.split()[1] == "hip" for ln in r.stdout.splitlines())


def test_host_only_build_of_the_package_without_the_hip_library(tmp_path):
    """./configure's second branch (no librcppsparse_hip.so on the machine): the package's sources plus
    src/nohip_stubs.c, NOT linked against the library.  It behaves like the GPU build on a box without a GPU:
    columnSums answers on the host with the reference's bits, gpuMatrix is an R error."""
    rpkg = os.path.join(ROOT, "rcppsparse_amd", "host", "rpkg")
    subprocess.run(["bash", os.path.join(rpkg, "assemble.sh")], check=True, stdout=subprocess.DEVNULL)
    mock = os.path.join(ROOT, "tests", "mock_rcpp")
    stubs = tmp_path / "stubs.o"
    subprocess.run(["gcc", "-c", "-O1", "-Wall", "-o", str(stubs), os.path.join(rpkg, "src", "nohip_stubs.c")], check=True)
    exe = str(tmp_path / "driver_nohip")
    subprocess.run(["g++", "-std=c++14", "-O1", "-Wall", "-I", mock, "-o", exe, os.path.join(mock, "driver.cpp"),
                    os.path.join(rpkg, "src", "columnSums.cpp"), os.path.join(rpkg, "src", "gpuMatrix.cpp"),
                    os.path.join(rpkg, "src", "rcpp_glue.cpp"), str(stubs)], check=True)
    libs = subprocess.run(["ldd", exe], capture_output=True, text=True, check=True).stdout
    assert "rcppsparse_hip" not in libs and "amdhip" not in libs
    r = run(exe, "kat_cpu", require_gpu=None)
    assert r.returncode == 0 and r.stdout.splitlines()[0].split() == KAT_HEX
    r = run(exe, "handle_nogpu")
    assert r.returncode == 0 and "no HIP device" in r.stdout
    # the configure script picks that branch by itself when the library is not where it looks
    work = tmp_path / "pkg"
    subprocess.run(["cp", "-r", rpkg, str(work)], check=True)
    env = dict(os.environ, RCPPSPARSE_HIP_LIB=str(tmp_path / "nowhere"))
    out = subprocess.run(["sh", "./configure"], cwd=str(work), env=env, capture_output=True, text=True, check=True).stdout
    def makevars():
        return dict(ln.split(" = ", 1) for ln in (work / "src" / "Makevars").read_text().splitlines()
                    if " = " in ln and not ln.startswith("#"))
    mk = makevars()
    assert "host-only build" in out and "nohip_stubs.o" in mk["OBJECTS"] and "-lrcppsparse_hip" not in mk["PKG_LIBS"]
    env["RCPPSPARSE_HIP_LIB"] = os.path.join(ROOT, "rcppsparse_amd")
    subprocess.run(["sh", "./configure"], cwd=str(work), env=env, check=True, stdout=subprocess.DEVNULL)
    mk = makevars()
    assert "-lrcppsparse_hip" in mk["PKG_LIBS"] and "nohip_stubs.o" not in mk["OBJECTS"]


def test_nothing_in_the_product_tree_touches_the_oracle():
    """The host answer is the product's own loop over its own InnerIterator -- not the oracle.  No file under
    rcppsparse_amd/ includes, imports, loads or links anything of oracle/ (comments may name it)."""
    import re
    pat = re.compile(r'#\s*include\s*[<"][^>"]*oracle|import\s+oracle|from\s+oracle|liboracle|oracle/|oracle\.(build|column_sums|gen_)')
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "rcppsparse_amd")):
        if "__pycache__" in base or os.sep + "build" in base:
            continue
        for f in files:
            if f.endswith((".so", ".o", ".pyc")):
                continue
            path = os.path.join(base, f)
            for n, line in enumerate(open(path, errors="replace"), 1):
                code = line.split("//")[0].split("#")[0] if not line.lstrip().startswith(("#include", "# include")) else line
                if pat.search(code):
                    bad.append(f"{os.path.relpath(path, ROOT)}:{n}: {line.strip()}")
    assert not bad, bad
    for lib in ("librcppsparse_hip.so", "librcppsparse_host.so"):
        libs = subprocess.run(["ldd", os.path.join(ROOT, "rcppsparse_amd", lib)], capture_output=True, text=True).stdout
        assert "oracle" not in libs


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
    # with a GPU present a matrix large enough goes to the device ("now"); the vignette's five entries are below the
    # offload threshold, so without a required GPU the host loop answered them (round 5; RcppSparse.min_nnz)
    assert run(driver, "backend", require_gpu=None).stdout.split() == ["hip", "none"]
    r = run(driver, "kat_cpu", require_gpu=None)
    assert "backend now=hip last=cpu" in r.stdout
    env = dict(os.environ, RCPPSPARSE_MIN_NNZ="0")
    env.pop("RCPPSPARSE_REQUIRE_GPU", None)
    r = subprocess.run([driver, "kat_cpu"], capture_output=True, text=True, timeout=120, env=env)
    assert "backend now=hip last=hip" in r.stdout          # threshold 0: every matrix to the device


@pytest.mark.gpu
def test_gpu_matrix_handles_are_told_apart_by_tag_and_sized_by_the_native_handle(driver):
    """dispatch relied on the R-level class attribute and the output length on the mutable Dim
    attribute.  Now the external pointer's tag decides and the native handle gives the sizes."""
    r = run(driver, "handle_swap")
    assert r.returncode == 0, (r.returncode, r.stdout + r.stderr)
    assert "handle swap ok" in r.stdout
