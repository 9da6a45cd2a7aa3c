#!/usr/bin/env python3
"""Soak of the slice-major row-restricted sums (colsums_rowslices.hip) against the oracle's restricted loop:
random row counts above 2^20 (2..7 slices, partial last slice), 1..50000 columns (the form is forced: rsp_debug_set("row_slices", 2)), Poisson column lengths from 0..5 to hundreds
with empty and long columns (some beyond the device-side guard, which must hand the call to the general kernel),
row sets from empty to full, both restrictions.  Prints one JSON line.
    python3 tools/soak_row_slices.py [cases] [seed]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import oracle
from rcppsparse_amd import capi, synth

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
capi.load()
capi.set_row_slices(2)          # the form wherever it is possible: the shapes here are far smaller than the ones it is chosen for
S = 1 << 20
worst, forms, t0 = 0.0, {}, time.time()
for case in range(cases):
    rng = np.random.default_rng(seed0 * 100_000 + case)
    nsl = int(rng.integers(2, 8))
    nrow = (nsl - 1) * S + int(rng.integers(1, S + 1))
    ncol = int(rng.integers(1, 300)) if case % 4 == 1 else int(rng.integers(300, 50_000))
    mean = int(rng.integers(1, 60)) * nsl if case % 4 != 3 else int(rng.integers(0, 6))
    counts = rng.poisson(mean, size=ncol).astype(np.int64)
    counts[rng.random(ncol) < 0.02] = 0
    for _ in range(int(rng.integers(0, 4))):                         # long columns, some beyond the guard (16 x mean + 4096)
        counts[int(rng.integers(0, ncol))] = int(rng.integers(mean * 4, mean * 40 + 20_000))
    col = np.repeat(np.arange(ncol, dtype=np.int64), counts)
    if case % 3 == 0:                                                # rows crowded into one slice
        row = rng.integers(0, min(nrow, S + 5000), size=col.size, dtype=np.int64)
    else:
        row = rng.integers(0, nrow, size=col.size, dtype=np.int64)
    key = np.unique(col * nrow + row)
    col, row = key // nrow, (key % nrow).astype(np.int32)
    p = np.zeros(ncol + 1, dtype=np.int64)
    np.add.at(p, col + 1, 1)
    p = np.cumsum(p).astype(np.int32)
    nnz = int(p[-1])
    if nnz == 0:
        continue
    x = synth.gen_values(nnz, seed=case, kind=case % 2)
    dens = [0.0, 1.0, 0.5, 0.05, 0.95][case % 5]
    bits = capi.row_set_bitmap(np.flatnonzero(rng.random(nrow) < dens), nrow)
    form = capi.in_rows_form(nrow, ncol, nnz)
    forms[form] = forms.get(form, 0) + 1
    xt, it, pt, bt = (torch.from_numpy(a).cuda() for a in (x, row, p, bits))
    for comp in (False, True):
        got = capi.column_sums_in_rows_device(xt, it, pt, nrow, bt, comp).cpu().numpy()
        ref = oracle.column_sums_in_rows(x, row, p, bits, comp)
        keep = (((bits[row >> 5] >> (row & 31).astype(np.uint32)) & 1) == 1) != comp
        scale = oracle.column_abs_sums(np.where(keep, x, 0.0), p)
        err = np.abs(got - ref)
        bad = err > 1e-12 * scale
        if bad.any():
            c = int(np.flatnonzero(bad)[0])
            print(json.dumps({"FAILED": case, "column": c, "got": float(got[c]), "ref": float(ref[c]), "nrow": nrow,
                              "ncol": ncol, "complement": comp, "form": form}))
            sys.exit(1)
        worst = max(worst, float(np.max(err / np.maximum(scale, 1e-300))))
    if case % 10 == 9:
        print(f"# {case + 1} cases, {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
print(json.dumps({"cases": cases, "seed": seed0, "forms": forms, "max_err_over_l1": worst,
                  "seconds": round(time.time() - t0, 1)}))
