"""The C++ host mirror of the reference interface (rcppsparse_amd/host), driven
through its Rcpp-free test seam.  CPU tests cover the class semantics the drop-in
must keep (reference inst/include/RcppSparse.h:25-396); the gpu test calls the
exported columnSums(Matrix&) exactly as the Rcpp glue would."""
import numpy as np
import pytest
import scipy.sparse as sp

import oracle
from conftest import golden_names, load_golden
from rcppsparse_amd import hostseam, synth


def dense_of(m):
    return sp.csc_matrix((m["x"], m["i"], m["p"]), shape=tuple(int(v) for v in m["Dim"])).toarray()


@pytest.fixture(scope="module")
def mat():
    return synth.rsparsematrix(40, 30, density=0.15, seed=21)


def test_sizes_and_accessors(mat):
    s = hostseam.sizes(mat, col0=3)
    assert s["rows"] == s["nrow"] == 40 and s["cols"] == s["ncol"] == 30
    assert s["n_nonzero"] == mat["x"].size
    assert s["InnerNNZs"] == mat["p"][4] - mat["p"][3]


def test_inner_iterator_walks_exactly_the_storage_range(mat):
    # RcppSparse.h:218-233: [p[col], p[col+1]), value()=x[index], row()=i[index], col()=col
    for col in range(30):
        rows, vals, cols = hostseam.walk_column(mat, col)
        lo, hi = mat["p"][col], mat["p"][col + 1]
        assert np.array_equal(rows, mat["i"][lo:hi])
        assert vals.tobytes() == mat["x"][lo:hi].tobytes()
        assert np.all(cols == col) and len(rows) == hi - lo


@pytest.mark.parametrize("name", golden_names())
def test_reference_loop_over_the_mirror_matches_oracle_bits(name):
    # the reference's own double loop (example.cpp:28-30) compiled against the mirror class
    g = load_golden(name)
    got = hostseam.columnSums_by_iterator(g)
    assert got.tobytes() == g["sums"].tobytes()
    assert hostseam.dense(g, "colSums").tobytes() == g["sums"].tobytes()     # RcppSparse.h:131-137


def test_reductions_follow_reference_arithmetic(mat):
    assert hostseam.dense(mat, "rowSums").tobytes() == oracle.row_sums(mat["x"], mat["i"], mat["p"], 40).tobytes()
    assert hostseam.dense(mat, "colMeans").tobytes() == oracle.col_means(mat["x"], mat["p"], 40).tobytes()
    assert hostseam.dense(mat, "rowMeans").tobytes() == oracle.row_means(mat["x"], mat["i"], mat["p"], 40).tobytes()


def test_element_and_dense_views(mat):
    D = dense_of(mat)
    for r, c in [(0, 0), (5, 7), (39, 29), (12, 3)]:
        assert hostseam.dense(mat, "at", r, c)[0] == D[r, c]
    assert np.array_equal(hostseam.dense(mat, "col", 4), D[:, 4])
    assert np.array_equal(hostseam.dense(mat, "row", 9), D[9, :])
    assert np.allclose(hostseam.dense(mat, "crossprod"), D.T @ D, rtol=1e-13, atol=1e-13)


def test_subview_clones_and_index_helpers(mat):
    # RcppSparse.h:73-128 (operator() overloads, col / row of several indices, operator[]) and
    # :198-215 (InnerIndices / emptyInnerIndices)
    D = dense_of(mat)
    rows, cols = [0, 7, 7, 39, 3], [29, 0, 12]
    assert np.array_equal(hostseam.subview(mat, "row_cols", cols=cols, a0=7), D[7, cols])
    assert np.array_equal(hostseam.subview(mat, "rows_col", rows=rows, a0=12), D[rows, 12])
    assert np.array_equal(hostseam.subview(mat, "rows_cols", rows=rows, cols=cols), D[np.ix_(rows, cols)])
    assert np.array_equal(hostseam.subview(mat, "cols", cols=cols), D[:, cols])
    assert np.array_equal(hostseam.subview(mat, "rows", rows=rows), D[rows, :])
    assert hostseam.subview(mat, "linear", a0=5)[0] == mat["x"][5]
    for c in (0, 11, 29):
        lo, hi = mat["p"][c], mat["p"][c + 1]
        nz = mat["i"][lo:hi]
        assert np.array_equal(hostseam.subview(mat, "InnerIndices", a0=c), nz)
        assert np.array_equal(hostseam.subview(mat, "emptyInnerIndices", a0=c), np.setdiff1d(np.arange(40), nz))


def test_restricted_iterators_implement_documented_intent(mat):
    # InnerIteratorInRange / NotInRange: column entries whose row is / is not in the sorted set
    rng = np.random.default_rng(0)
    for col in range(0, 30, 3):
        s = np.sort(rng.choice(40, size=12, replace=False)).astype(np.uint32)
        lo, hi = mat["p"][col], mat["p"][col + 1]
        rows, vals = mat["i"][lo:hi], mat["x"][lo:hi]
        keep = np.isin(rows, s)
        r_in, v_in = hostseam.walk_restricted(mat, col, s, "in")
        assert np.array_equal(r_in, rows[keep]) and np.array_equal(v_in, vals[keep])
        r_out, v_out = hostseam.walk_restricted(mat, col, s, "not_in")
        assert np.array_equal(r_out, rows[~keep]) and np.array_equal(v_out, vals[~keep])
    # empty set / empty column corner cases (the reference reads out of bounds here)
    empty = np.array([], dtype=np.uint32)
    assert len(hostseam.walk_restricted(mat, 0, empty, "in")[0]) == 0
    lo, hi = mat["p"][0], mat["p"][1]
    assert np.array_equal(hostseam.walk_restricted(mat, 0, empty, "not_in")[0], mat["i"][lo:hi])


def test_row_iterator_and_symmetry(mat):
    D = dense_of(mat)
    for r in (0, 7, 39):
        cols, vals = hostseam.walk_restricted(mat, r, np.array([], dtype=np.uint32), "row")
        nz = np.flatnonzero(D[r, :] != 0)
        assert np.array_equal(cols, nz) and np.array_equal(vals, D[r, nz])
    assert not hostseam.is_appx_symmetric(mat)            # 40 x 30 is not square
    S = sp.random(25, 25, density=0.2, random_state=3, format="csc")
    S = (S + S.T).tocsc()
    S.sort_indices()
    sym = {"x": S.data, "i": S.indices, "p": S.indptr, "Dim": np.array([25, 25])}
    assert hostseam.is_appx_symmetric(sym)


def test_transpose_is_a_valid_csc_of_the_transpose(mat):
    T = hostseam.transpose(mat)
    assert np.array_equal(dense_of(T), dense_of(mat).T)
    for c in range(40):                                    # rows ascending inside each column
        seg = T["i"][T["p"][c]:T["p"][c + 1]]
        assert np.all(np.diff(seg) > 0)


def test_s4_constructor_checks_slots_with_the_reference_message():
    hostseam.construct_from_s4(0b1111)
    for mask in (0b0111, 0b1011, 0b1101, 0b1110, 0):
        with pytest.raises(hostseam.SeamError) as e:
            hostseam.construct_from_s4(mask)
        assert str(e.value) == "Cannot construct RcppSparse::Matrix from this S4 object"  # RcppSparse.h:36


def test_by_reference_semantics():
    assert hostseam.shares_storage()


@pytest.mark.parametrize("name", golden_names())
def test_exported_columnSums_answers_on_the_host_when_the_machine_has_no_gpu(name):
    """with zero HIP devices the exported columnSums(Matrix&) runs the reference's loop
    (example.cpp:28-30) over the mirror's own InnerIterator into the vector it has already allocated -- the
    reference's bits on every golden fixture -- and says so; with a GPU required (this suite's default, set in
    conftest.py) the same call stays the error it was."""
    from rcppsparse_amd import capi
    if capi.device_count() > 0:
        pytest.skip("a GPU is present: the host loop is never selected")
    g = load_golden(name)
    ncol = int(g["Dim"][1])
    if ncol > 0:                                     # (a matrix without columns has nothing to ask a device for)
        with pytest.raises(hostseam.SeamError) as e:
            hostseam.columnSums(g)                   # RCPPSPARSE_REQUIRE_GPU=1 from conftest
        assert "no HIP device" in str(e.value)
    assert hostseam.backend() == "none"
    got = hostseam.columnSums_opt(g, require_gpu=0)
    assert hostseam.backend(last=True) == "cpu" and hostseam.backend(require_gpu=0) == "cpu"
    assert got.tobytes() == np.asarray(g["sums"], dtype=np.float64).tobytes()
    if ncol > 0:
        with pytest.raises(hostseam.SeamError):
            hostseam.columnSums_opt(g, require_gpu=1)
        assert hostseam.backend(last=True) == "none"    # the failed call answered nothing


@pytest.mark.gpu
def test_with_a_gpu_present_the_host_loop_answers_small_matrices_only_when_allowed():
    """Round 5 ("min-nnz threshold for GPU offload"): with a GPU present the exported columnSums goes to the
    device -- except for matrices below the threshold when no GPU is REQUIRED, which the reference's own loop answers faster
    (profiles/r05_one_shot.json).  This suite runs with RCPPSPARSE_REQUIRE_GPU=1 (tests/conftest.py), so require_gpu = -1
    (the environment decides) and 1 mean the device whatever the size; 0 lets the threshold speak."""
    g = load_golden(golden_names()[0])
    nnz = int(len(g["x"]))
    for req in (-1, 1):
        assert hostseam.backend(require_gpu=req) == "hip"
        hostseam.columnSums_opt(g, require_gpu=req)
        assert hostseam.backend(last=True) == "hip"
        assert hostseam.backend_for(nnz, require_gpu=req, min_nnz=10**9) == "hip"
    assert hostseam.backend(require_gpu=0) == "hip"                    # (a matrix large enough)
    a = hostseam.columnSums_opt2(g, require_gpu=0, min_nnz=nnz + 1)     # below the threshold: the host loop
    assert hostseam.backend(last=True) == "cpu"
    b = hostseam.columnSums_opt2(g, require_gpu=0, min_nnz=nnz)         # at it: the device
    assert hostseam.backend(last=True) == "hip"
    c = hostseam.columnSums_opt2(g, require_gpu=0, min_nnz=0)
    assert hostseam.backend(last=True) == "hip"
    assert a.tobytes() == np.asarray(g["sums"], dtype=np.float64).tobytes() and b.tobytes() == c.tobytes()
    scale = oracle.column_abs_sums(g["x"], g["p"])
    assert np.all(np.abs(a - b) <= 1e-12 * scale)
    hostseam.columnSums_opt(g, require_gpu=0)                           # the default threshold (250000): a golden fixture is small
    assert hostseam.backend(last=True) == "cpu"


@pytest.mark.gpu
def test_exported_columnSums_over_several_devices_when_asked(monkeypatch):
    """RCPPSPARSE_DEVICES (round 5, opt-in): the exported columnSums spreads a call on host data over several GPUs --
    rsp_column_sums_host_multi: nnz-balanced column ranges, one host thread and one host link per range.  On this box the
    "several" are device 0 three times (an ordinal may repeat); "all" is every visible device.  Same sums as the
    one-device call within the tolerance, bit-identical run to run; not set: one device as before."""
    m = synth.rsparsematrix(4000, 1500, density=0.05, seed=31)
    monkeypatch.delenv("RCPPSPARSE_DEVICES", raising=False)
    one = hostseam.columnSums(m)
    scale = oracle.column_abs_sums(m["x"], m["p"])
    ref = oracle.column_sums(m["x"], m["p"])
    assert np.all(np.abs(one - ref) <= 1e-12 * scale)
    for setting in ("0,0,0", "all", " 0 , 0"):
        monkeypatch.setenv("RCPPSPARSE_DEVICES", setting)
        got = hostseam.columnSums(m)
        assert hostseam.backend(last=True) == "hip"
        assert np.all(np.abs(got - ref) <= 1e-12 * scale), setting
        assert got.tobytes() == hostseam.columnSums(m).tobytes()
    monkeypatch.setenv("RCPPSPARSE_DEVICES", "7")                       # no such device on a one-GPU box: an R error, not a crash
    from rcppsparse_amd import capi
    if capi.device_count() < 8:
        with pytest.raises(hostseam.SeamError) as e:
            hostseam.columnSums(m)
        assert "out of range" in str(e.value)


@pytest.mark.gpu
@pytest.mark.parametrize("name", golden_names())
def test_exported_columnSums_through_the_hip_shim(name):
    g = load_golden(name)
    got = hostseam.columnSums(g)
    ref = g["sums"]
    scale = oracle.column_abs_sums(g["x"], g["p"])
    ok = np.isfinite(ref)
    assert got.shape == ref.shape
    assert np.all(np.abs(got[ok] - ref[ok]) <= 1e-12 * scale[ok])
    assert np.array_equal(np.isnan(got), np.isnan(ref))


@pytest.mark.gpu
def test_exported_columnSums_readme_example():
    # reference README.md:33-38: A <- rsparsematrix(10, 10, 0.1); columnSums(A)
    A = synth.rsparsematrix(10, 10, density=0.1, seed=1)
    got = hostseam.columnSums(A)
    assert got.shape == (10,) and got.tobytes() == oracle.column_sums(A["x"], A["p"]).tobytes()


@pytest.mark.gpu
def test_plain_c_caller_of_the_abi(tmp_path):
    """A C program (gcc, no Python/torch/C++ in the process) drives the C ABI: one-shot, handle,
    means, rowSums, error status.  Also proves the library runs on the system HIP runtime."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "rcppsparse_amd")
    exe = str(tmp_path / "cabi_smoke")
    subprocess.run(["gcc", "-O1", "-o", exe, os.path.join(root, "tests", "c", "cabi_smoke.c"),
                    "-I", os.path.join(root, "include"), "-L", libdir, "-lrcppsparse_hip", "-lm",
                    f"-Wl,-rpath,{libdir}"], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "cabi_smoke ok" in r.stdout


def test_host_mirror_under_address_and_ub_sanitizers(tmp_path):
    """rcppsparse_core.hpp through the seam, 1500 random small matrices incl. empty columns, empty
    row sets and single rows/columns, in an ASan + UBSan build (CPU entry points only)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "rcppsparse_amd")
    hostseam.load()                                   # makes sure librcppsparse_hip.so exists
    exe = str(tmp_path / "host_selftest")
    subprocess.run(["g++", "-std=c++14", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                    "-fno-sanitize-recover=all", os.path.join(root, "tests", "c", "host_selftest.cpp"),
                    os.path.join(libdir, "host", "host_seam.cpp"), "-o", exe, "-L", libdir, "-lrcppsparse_hip",
                    f"-Wl,-rpath,{libdir}"], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert r.returncode == 0, r.stdout + r.stderr[-2000:]
    assert "host selftest ok" in r.stdout


@pytest.mark.gpu
def test_in_a_forked_child_the_exported_columnSums_answers_on_the_host():
    """R's parallel::mclapply forks, and the HIP runtime does not survive a fork.  Round 6: a child of a process that has used
    the GPU is, to the C ABI, a machine without a device -- so the exported columnSums (the drop-in of reference
    src/example.cpp:26-32) runs the reference's own loop on the host there, with the reference's bits, instead of entering a
    runtime that may hang; with a GPU REQUIRED it is an R error.  The parent keeps its GPU."""
    import select
    m = synth.rsparsematrix(3000, 800, density=0.05, seed=77)
    want = oracle.column_sums(m["x"], m["p"])
    hostseam.columnSums_opt(m, require_gpu=1)
    assert hostseam.backend(last=True) == "hip"                          # the parent has used the GPU
    os_ = __import__("os")
    r, w = os_.pipe()
    pid = os_.fork()
    if pid == 0:
        os_.close(r)
        said = "?"
        try:
            got = hostseam.columnSums_opt2(m, require_gpu=0, min_nnz=0)  # (the threshold would send this matrix to the device)
            said = f"{hostseam.backend(last=True)};{got.tobytes() == want.tobytes()}"
            try:
                hostseam.columnSums_opt(m, require_gpu=1)
                said += ";answered"
            except Exception as e:   # noqa: BLE001
                said += ";error" if "GPU" in str(e) or "device" in str(e) else ";other " + str(e)[:60]
        except BaseException as e:   # noqa: BLE001
            said = "failed: " + repr(e)[:100]
        try:
            os_.write(w, said.encode())
        finally:
            os_._exit(0)
    os_.close(w)
    ready, _, _ = select.select([r], [], [], 30)
    assert ready, "the forked child did not answer within 30 s"
    said = os_.read(r, 400).decode()
    os_.close(r)
    os_.waitpid(pid, 0)
    assert said == "cpu;True;error", said
    hostseam.columnSums_opt(m, require_gpu=1)
    assert hostseam.backend(last=True) == "hip"                          # the parent still computes on its GPU
