"""CPU-only: pins the oracle (CPU restatement of reference src/example.cpp:26-32).

The reference holds no golden vectors for columnSums; the pins
are (1) the literal matrix of reference vignettes/Documentation.Rmd:213-216 with
hand-derived exact IEEE sums, (2) SciPy as an independent second opinion,
(3) a pure-Python second transcription of the same loop.
"""
import numpy as np
import pytest
import scipy.sparse as sp

import oracle
from conftest import golden_names, load_golden
from rcppsparse_amd import synth


def test_kat_vignette_exact_bits():
    # Documentation.Rmd:213-216: x, i, p, Dim literal; sequential += from +0.0
    x = [0.41, 0.35, 0.84, 0.37, 0.26]
    p = [0, 0, 1, 2, 4, 5]
    got = oracle.column_sums(x, p)
    want_hex = ["0x0.0p+0", "0x1.a3d70a3d70a3dp-2", "0x1.6666666666666p-2",
                "0x1.35c28f5c28f5cp+0", "0x1.0a3d70a3d70a4p-2"]
    assert [float.hex(float(v)) for v in got] == want_hex
    assert np.array_equal(got, np.array([0.0, 0.41, 0.35, 0.84 + 0.37, 0.26]))


@pytest.mark.parametrize("name", golden_names())
def test_golden_fixtures_bit_exact(name):
    g = load_golden(name)
    ncol = int(g["Dim"][1])
    got = oracle.column_sums(g["x"], g["p"], ncol, i=g["i"], nrow=int(g["Dim"][0]))
    assert got.shape == (ncol,)
    assert got.tobytes() == g["sums"].tobytes()          # bit-exact incl. NaN / signed zero
    # the iterator-free form RcppSparse.h:131-137 gives the same bits
    assert oracle.col_sums(g["x"], g["p"], ncol).tobytes() == g["sums"].tobytes()


@pytest.mark.parametrize("name", [n for n in golden_names() if "3000x700" not in n])
def test_python_transcription_agrees(name):
    g = load_golden(name)
    got = oracle.column_sums_py(g["x"], g["p"], int(g["Dim"][1]))
    if name == "na_payload":
        # columns 0-5 hold NA_real_ beside finite / infinite values: NA (payload 1954, quieted) from the C restatement and
        # from this transcription alike.  Columns 6-8 mix NA with a NaN of another payload: WHICH payload an x86 add returns
        # depends on the operand order the compiler chose -- gcc's loop and CPython's float add differ on this very box
        # (what R's documentation means by "platform-dependent") -- so only the class is common ground there.
        assert got[:6].tobytes() == g["sums"][:6].tobytes() and int(got[:1].view(np.uint64)[0]) == 0x7FF80000000007A2
        assert np.all(np.isnan(got[6:])) and np.all(np.isnan(g["sums"][6:]))
        return
    assert got.tobytes() == g["sums"].tobytes()


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_scipy_cross_check(seed):
    m = synth.rsparsematrix(400, 300, density=0.03, seed=seed)
    A = sp.csc_matrix((m["x"], m["i"], m["p"]), shape=(400, 300))
    ref = np.asarray(A.sum(axis=0)).ravel()
    got = oracle.column_sums(m["x"], m["p"])
    scale = np.maximum(oracle.column_abs_sums(m["x"], m["p"]), 1e-300)
    assert np.all(np.abs(got - ref) <= 1e-13 * scale)
    red = np.add.reduceat(np.append(m["x"], 0.0), np.minimum(m["p"][:-1], len(m["x"])))
    red[np.diff(m["p"]) == 0] = 0.0
    assert np.all(np.abs(got - red) <= 1e-13 * scale)


@pytest.mark.parametrize("seed,shape,density", [(21, (300, 40), 0.2), (22, (2000, 130), 0.05), (23, (64, 70), 0.6), (24, (500, 9), 0.0)])
def test_crossprod_restatement_against_scipy(seed, shape, density):
    """oracle.crossprod (the reference's sorted merges, RcppSparse.h:159-194) against SciPy's t(A) %*% A: the same numbers up
    to the order of the adds (1e-13 of sum |x1 x2|), exactly symmetric, exactly zero where two columns share no row.  (GPU
    tests of the wide matrix-core forms use SciPy's product directly where the merges would take seconds per case.)"""
    nrow, ncol = shape
    m = synth.rsparsematrix(nrow, ncol, density=density, seed=seed)
    x, i, p = m["x"], m["i"], m["p"]
    got = oracle.crossprod(x, i, p)
    A = sp.csc_matrix((x, i, p), shape=shape)
    B = sp.csc_matrix((np.abs(x), i, p), shape=shape)
    ref = np.asarray((A.T @ A).todense())
    scale = np.asarray((B.T @ B).todense())
    assert got.shape == (ncol, ncol) and np.array_equal(got, got.T)
    assert np.all(np.abs(got - ref) <= 1e-13 * scale)
    assert np.all(got[scale == 0] == 0)


def test_empty_columns_are_positive_zero():
    g = load_golden("stored_zeros")
    s = oracle.column_sums(g["x"], g["p"])
    assert not np.signbit(s[1]) and s[1] == 0.0     # column of -0.0 only -> +0.0
    assert not np.signbit(s[2]) and s[2] == 0.0     # empty column


def test_row_and_mean_variants():
    m = synth.rsparsematrix(50, 40, density=0.2, seed=5)
    A = sp.csc_matrix((m["x"], m["i"], m["p"]), shape=(50, 40)).toarray()
    assert np.allclose(oracle.row_sums(m["x"], m["i"], m["p"], 50), A.sum(axis=1), atol=1e-12)
    assert np.allclose(oracle.col_means(m["x"], m["p"], 50), A.sum(axis=0) / 50, atol=1e-13)
    assert np.allclose(oracle.row_means(m["x"], m["i"], m["p"], 50), A.sum(axis=1) / 40, atol=1e-13)
    # colMeans divides the sums (RcppSparse.h:147-148), bit for bit
    assert np.array_equal(oracle.col_means(m["x"], m["p"], 50), oracle.col_sums(m["x"], m["p"]) / 50)


@pytest.mark.parametrize("kind", [0, 1])
def test_generators_agree_bitwise(kind):
    a = oracle.gen_values(5000, seed=42, first_idx=123456789012, kind=kind)
    b = synth.gen_values(5000, seed=42, first_idx=123456789012, kind=kind)
    assert a.tobytes() == b.tobytes()
    if kind == 0:
        assert a.min() >= -5.1 and a.max() <= 5.1
        assert np.array_equal(a, np.round(a * 100) / 100.0)   # two decimals, correctly rounded
    else:
        assert a.min() >= 0.0 and a.max() < 1.0


def test_oracle_under_address_and_ub_sanitizers(tmp_path):
    """The oracle's C loops, KAT + 3000 random ragged matrices, built with ASan + UBSan
    (sanitizers run on the CPU side only: the GPU pool offers none)."""
    import os
    import subprocess
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "selftest")
    subprocess.run(["gcc", "-O1", "-g", "-std=c11", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                    "-fno-sanitize-recover=all", "-fno-fast-math", "-ffp-contract=off",
                    os.path.join(here, "oracle", "selftest.c"), os.path.join(here, "oracle", "colsums_oracle.c"),
                    "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "oracle selftest ok" in r.stdout


def test_threaded_bench_variant_is_bit_identical_to_the_serial_loop():
    # oracle/colsums_threads.c (bench.py's optional all-cores figure): same per-column loop
    p = synth.offsets_from_counts(synth.zipf_counts(3000, 400_000, seed=3, nrow=100_000))
    x = oracle.gen_values(int(p[-1]), 5, 0, 0)
    assert np.array_equal(oracle.gen_values_threads(x.size, 5, 0, 0, nthreads=4), x)
    ref = oracle.column_sums(x, p)
    for t in (1, 3, 8):
        assert oracle.column_sums_threads(x, p, t).tobytes() == ref.tobytes()
