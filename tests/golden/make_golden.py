"""Generates tests/golden/*.npz -- small seeded dgCMatrix fixtures + expected columnSums.

Run from the repo root:  python tests/golden/make_golden.py

The reference has no golden vectors for this path and it
cannot be built here (needs Rcpp/R), so expected values come from the CPU
restatement in oracle/ (reference src/example.cpp:26-32 restated) and are
cross-checked here against SciPy's csc_matrix.sum(axis=0) before being
written.  The one reference-derived fixture is `kat_vignette`: the literal 5x5
matrix of reference vignettes/Documentation.Rmd:213-216.

Fixtures hold data only: x, i, p, Dim, expected sums (float64, exact bits).
"""
import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from rcppsparse_amd import synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def save(name, x, i, p, dim):
    x = np.asarray(x, dtype=np.float64)
    i = np.asarray(i, dtype=np.int32)
    p = np.asarray(p, dtype=np.int32)
    dim = np.asarray(dim, dtype=np.int32)
    sums = oracle.column_sums(x, p, dim[1], i=i, nrow=dim[0])
    # second opinion (different summation order is possible -> tolerance, not bits)
    if x.size and dim[1] > 0 and np.all(np.isfinite(x)):
        A = sp.csc_matrix((x, i, p), shape=tuple(int(d) for d in dim))
        ref2 = np.asarray(A.sum(axis=0)).ravel()
        scale = np.maximum(oracle.column_abs_sums(x, p, dim[1]), 1e-300)
        assert np.all(np.abs(ref2 - sums) <= 1e-13 * scale), name
    np.savez(os.path.join(OUT, name + ".npz"), x=x, i=i, p=p, Dim=dim, sums=sums)
    print(f"{name}: {dim[0]}x{dim[1]} nnz={x.size}")


def main():
    # reference vignettes/Documentation.Rmd:213-216
    save("kat_vignette", [0.41, 0.35, 0.84, 0.37, 0.26], [0, 2, 0, 1, 1], [0, 0, 1, 2, 4, 5], [5, 5])

    # BASELINE config C1: rsparsematrix(10, 10, 0.1)  (reference README.md:33-38)
    m = synth.rsparsematrix(10, 10, density=0.1, seed=1)
    save("c1_10x10_d0.1", m["x"], m["i"], m["p"], m["Dim"])
    # man page example shape: rsparsematrix(10, 5, 0.5)  (reference src/example.cpp:10)
    m = synth.rsparsematrix(10, 5, density=0.5, seed=2)
    save("man_10x5_d0.5", m["x"], m["i"], m["p"], m["Dim"])
    # vignette shape: rsparsematrix(5, 5, 0.5)  (reference Documentation.Rmd:315)
    m = synth.rsparsematrix(5, 5, density=0.5, seed=3)
    save("vignette_5x5_d0.5", m["x"], m["i"], m["p"], m["Dim"])

    m = synth.rsparsematrix(1000, 200, density=0.05, seed=4)
    save("uniform_1000x200_d0.05", m["x"], m["i"], m["p"], m["Dim"])
    m = synth.rsparsematrix(3000, 700, density=0.02, seed=5, kind=1)
    save("positive_3000x700_d0.02", m["x"], m["i"], m["p"], m["Dim"])

    # all columns empty
    save("all_empty_7cols", [], [], [0] * 8, [9, 7])
    # zero columns
    save("zero_cols", [], [], [0], [4, 0])
    # one column holds every nonzero, surrounded by empty columns
    n = 5000
    save("single_dense_column", synth.gen_values(n, 6), np.arange(n), [0, 0, 0, n, n, n], [n, 5])
    # signed cancellation: +big, small, -big in each column
    big = 1e15
    x = np.array([big, 1.0, -big, big, -big, 3.0, 1e-3, -1e-3, 1e-3], dtype=np.float64)
    save("cancellation", x, [0, 1, 2, 0, 1, 2, 0, 1, 2], [0, 3, 6, 9], [3, 3])
    # explicit stored zeros and negative zeros (empty / -0.0 columns must come out +0.0)
    x = np.array([0.0, -0.0, -0.0, -0.0, 2.5, -0.0], dtype=np.float64)
    save("stored_zeros", x, [0, 1, 0, 1, 0, 1], [0, 2, 4, 4, 6], [2, 4])
    # non-finite values propagate like the plain += of the reference
    x = np.array([1.0, np.inf, 2.0, -np.inf, np.inf, np.nan, 1.0, 4.0], dtype=np.float64)
    save("nonfinite", x, [0, 1, 0, 1, 2, 0, 1, 0], [0, 2, 5, 7, 8], [3, 4])
    # R's NA_real_ is the signalling NaN 0x7FF00000000007A2 (low word 1954); the reference's plain += (src/example.cpp:30)
    # hands it back quieted with the payload kept, which R still prints as NA.  Columns 0-5 hold NA beside finite or
    # infinite values only: every form has to return the oracle's x86 bits for them (0x7FF80000000007A2).  Columns 6-8 mix
    # NA with a NaN of another payload: which of the two comes back depends on the order of the adds (R's own
    # documentation calls it platform-dependent); the x86 bits of the sequential loop are stored beside them.
    na = np.frombuffer(np.array([0x7FF00000000007A2], dtype=np.uint64).tobytes(), dtype=np.float64)[0]
    nan = np.frombuffer(np.array([0x7FF8000000000000], dtype=np.uint64).tobytes(), dtype=np.float64)[0]
    cols = [[na], [1.5, na, 2.5], [na, 1.5, 2.5], [1.5, 2.5, na], [na, np.inf], [-np.inf, na, 3.0],
            [nan, na], [na, nan], [np.inf, -np.inf, na]]
    x = np.concatenate([np.array(c, dtype=np.float64) for c in cols])
    assert x.view(np.uint64)[0] == 0x7FF00000000007A2          # the signalling pattern survived numpy
    p = np.concatenate([[0], np.cumsum([len(c) for c in cols])])
    i = np.concatenate([np.arange(len(c)) for c in cols])
    save("na_payload", x, i, p, [3, len(cols)])
    # ragged: lengths 0..40 interleaved with empties, odd nnz
    rng = np.random.default_rng(7)
    counts = rng.integers(0, 41, size=301)
    counts[::7] = 0
    p = synth.offsets_from_counts(counts)
    if p[-1] % 2 == 0:
        counts[1] += 1
        p = synth.offsets_from_counts(counts)
    x = synth.gen_values(int(p[-1]), 8)
    i = synth.row_indices(p, 64, 8)
    save("ragged_odd_nnz", x, i, p, [64, 301])


if __name__ == "__main__":
    main()
