#!/usr/bin/env python3
"""Long fuzz of the column-sum kernel against the oracle (a soak run, not part of the test suite):
random concatenations of column-length stretches (the generator of tests/test_gpu_parity.py plus
medium-length stretches), random chunk sizes, sums / max / min / sum of squares, for a given
number of seconds.  Prints one progress line every 50 cases; exits non-zero on the first mismatch.

    python tools/soak_fuzz.py --seconds 240 [--seed 0]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import oracle
from rcppsparse_amd import capi, synth
import test_gpu_parity
from test_gpu_parity import _random_structure, assert_parity, dev_colsums


def structure(rng):
    parts = [_random_structure(rng)]
    for _ in range(int(rng.integers(0, 4))):      # extra medium-length stretches (dense path, 2-8 lanes/column)
        mean = int(rng.choice([12, 18, 25, 40, 60, 90, 130, 200]))
        parts.append(rng.poisson(mean, size=int(rng.integers(20, 1500))))
        if rng.integers(0, 3) == 0:
            parts.append(np.zeros(int(rng.integers(1, 40)), dtype=np.int64))
    rng.shuffle(parts)
    return np.concatenate(parts).astype(np.int64)


def soak_next_rows(a):
    """crossprod (both kernels, bit-exact), rowSums and row-restricted column sums on random
    small matrices with row indices."""
    capi.set_crossprod_exact(True)   # (this soak is about the bit-identical kernels; tools/soak_crossprod_tall.py has the other)
    t0 = time.time()
    n = 0
    while time.time() - t0 < a.seconds:
        rng = np.random.default_rng(a.seed * 1_000_003 + n)
        nrow = int(rng.choice([1, 2, 7, 33, 64, 65, 200, 1000, 5000]))
        ncol = int(rng.choice([1, 2, 9, 63, 64, 65, 130, 300]))
        density = float(rng.choice([0.0, 0.003, 0.02, 0.1, 0.4, 0.9, 1.0]))
        m = synth.rsparsematrix(nrow, ncol, density=density, seed=n, kind=int(rng.integers(0, 2)))
        x, i, p = m["x"], m["i"], m["p"]
        xt, it, pt = torch.from_numpy(x).cuda(), torch.from_numpy(i).cuda(), torch.from_numpy(p).cuda()
        if x.size == 0:
            xt = torch.zeros(2, dtype=torch.float64, device="cuda")[:0]
            it = torch.zeros(2, dtype=torch.int32, device="cuda")[:0]
        what = f"case {n}: {nrow}x{ncol} density {density}"
        ref = oracle.crossprod(x, i, p)
        for tiles in (False, True):
            got = capi.crossprod_device(xt, it, pt, nrow, tiles=tiles).cpu().numpy().T
            assert np.array_equal(got, ref), (what, "crossprod", tiles)
        if x.size:
            got = capi.row_sums_device(xt, it, nrow).cpu().numpy()
            ref = oracle.row_sums(x, i, p, nrow)
            scale = oracle.row_sums(np.abs(x), i, p, nrow)
            assert np.all(np.abs(got - ref) <= 1e-12 * scale), (what, "rowSums")
            rows = np.flatnonzero(rng.random(nrow) < rng.random())
            bm = capi.row_set_bitmap(rows, nrow)
            for comp in (False, True):
                got = capi.column_sums_in_rows_device(xt, it, pt, nrow, torch.from_numpy(bm).cuda(), comp).cpu().numpy()
                ref = oracle.column_sums_in_rows(x, i, p, bm, comp)
                scale = oracle.column_abs_sums(x, p)
                assert np.all(np.abs(got - ref) <= 1e-12 * scale), (what, "in_rows", comp)
        if n % 25 == 0:
            # row-restricted sums with a bitmap that lives in LDS (shared by 16 / 8 / 6 wavefronts) or in L2:
            # many rows, random column structure
            nrow2 = int(rng.choice([140_000, 400_000, 700_000, 1_000_000, 1_048_576, 1_500_000]))
            counts = structure(rng)
            counts = np.minimum(counts, nrow2)
            p2 = synth.offsets_from_counts(counts)
            nnz2 = int(p2[-1])
            if 0 < nnz2 < 3_000_000:
                x2 = oracle.gen_values(nnz2, n, 0, 0)
                i2 = oracle.gen_row_indices(p2, nrow2, n)
                rows2 = np.flatnonzero(rng.random(nrow2) < rng.random())
                bm2 = capi.row_set_bitmap(rows2, nrow2)
                x2t, i2t, p2t = torch.from_numpy(x2).cuda(), torch.from_numpy(i2).cuda(), torch.from_numpy(p2).cuda()
                for comp in (False, True):
                    got = capi.column_sums_in_rows_device(x2t, i2t, p2t, nrow2, torch.from_numpy(bm2).cuda(), comp).cpu().numpy()
                    ref = oracle.column_sums_in_rows(x2, i2, p2, bm2, comp)
                    scale = oracle.column_abs_sums(x2, p2)
                    assert np.all(np.abs(got - ref) <= 1e-12 * scale), (n, "in_rows, many rows", nrow2, comp)
        n += 1
        if n % 50 == 0:
            print(f"{n} cases ok, {time.time() - t0:.0f} s", flush=True)
    print(f"soak ok: {n} cases in {time.time() - t0:.0f} s", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--what", default="colsums", choices=["colsums", "next"])
    ap.add_argument("--mode", default="general", choices=["general", "planned", "mixed"],
                    help="colsums: rsp_column_sums_device, the inspector-executor form (a third of the cases with every "
                         "column clipped to 64 entries so that the lean form applies, a third with the lean form "
                         "switched off), or both alternating")
    a = ap.parse_args()
    capi.load()
    forms = {}
    if a.what == "next":
        return soak_next_rows(a)
    t0 = time.time()
    n = 0
    while time.time() - t0 < a.seconds:
        rng = np.random.default_rng(a.seed * 1_000_003 + n)
        counts = structure(rng)
        planned = a.mode == "planned" or (a.mode == "mixed" and n % 2 == 1)
        lean_case = planned and n % 3 == 0
        if lean_case:
            counts = np.minimum(counts, 64)
        elif planned and n % 5 == 4 and counts.size >= 128:   # every column long: the columns form (one workgroup per column)
            counts = 2048 + counts[:int(rng.integers(128, 700))] % int(rng.integers(1, 6000))
        test_gpu_parity._MODE["launch"] = "planned" if planned else "general"
        capi.set_lean(not (planned and n % 3 == 1))
        p = synth.offsets_from_counts(counts)
        nnz = int(p[-1])
        kind = int(rng.integers(0, 2))
        x = synth.gen_values(nnz, seed=n, kind=kind)
        rows = int(rng.choice([0, 0, 1, 2, 3, 5, 8, 13, 16, 31, 64, 200]))
        capi.set_tuning(rows)
        try:
            before = dict(test_gpu_parity.PLANS_SEEN)
            got = dev_colsums(torch, x, p)
            assert_parity(got, x, p, positive=(kind == 1))
            for k, v in test_gpu_parity.PLANS_SEEN.items():
                if k != "n" and v != before.get(k, 0):
                    forms[k] = forms.get(k, 0) + 1
                    if k == "lean":      # the lean form's promise: the reference's bits in every column
                        assert got.tobytes() == oracle.column_sums(x, p).tobytes(), (n, "lean form is not bit-identical")
            if n % 4 == 0 and nnz > 0:               # the other combine policies on the same structure
                xt, pt = torch.from_numpy(x).cuda(), torch.from_numpy(p).cuda()
                for op in (capi.OP_MAX, capi.OP_MIN, capi.OP_SUM_SQUARES):
                    g = capi.column_reduce_device(xt, pt, op).cpu().numpy()
                    ref = oracle.column_reduce(x, p, op)
                    if op == capi.OP_SUM_SQUARES:
                        assert np.all(np.abs(g - ref) <= 1e-12 * np.maximum(ref, 1e-300)), (n, op)
                    else:
                        assert np.array_equal(g, ref), (n, op)
        except AssertionError:
            print(f"MISMATCH at case {n} (seed {a.seed}, chunk_rows {rows}, ncol {counts.size}, nnz {nnz})", flush=True)
            raise
        finally:
            capi.set_tuning(0)
        if n % 25 == 0:
            # row-restricted sums with a bitmap that lives in LDS (shared by 16 / 8 / 6 wavefronts) or in L2:
            # many rows, random column structure
            nrow2 = int(rng.choice([140_000, 400_000, 700_000, 1_000_000, 1_048_576, 1_500_000]))
            counts = structure(rng)
            counts = np.minimum(counts, nrow2)
            p2 = synth.offsets_from_counts(counts)
            nnz2 = int(p2[-1])
            if 0 < nnz2 < 3_000_000:
                x2 = oracle.gen_values(nnz2, n, 0, 0)
                i2 = oracle.gen_row_indices(p2, nrow2, n)
                rows2 = np.flatnonzero(rng.random(nrow2) < rng.random())
                bm2 = capi.row_set_bitmap(rows2, nrow2)
                x2t, i2t, p2t = torch.from_numpy(x2).cuda(), torch.from_numpy(i2).cuda(), torch.from_numpy(p2).cuda()
                for comp in (False, True):
                    got = capi.column_sums_in_rows_device(x2t, i2t, p2t, nrow2, torch.from_numpy(bm2).cuda(), comp).cpu().numpy()
                    ref = oracle.column_sums_in_rows(x2, i2, p2, bm2, comp)
                    scale = oracle.column_abs_sums(x2, p2)
                    assert np.all(np.abs(got - ref) <= 1e-12 * scale), (n, "in_rows, many rows", nrow2, comp)
        n += 1
        if n % 50 == 0:
            print(f"{n} cases ok, {time.time() - t0:.0f} s", flush=True)
    test_gpu_parity._MODE["launch"] = "general"
    capi.set_lean(True)
    print(f"soak ok: {n} cases in {time.time() - t0:.0f} s; plan forms seen: {forms}", flush=True)


if __name__ == "__main__":
    main()
