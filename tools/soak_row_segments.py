#!/usr/bin/env python3
"""Soak of the segments form of a handle's row sums (rowsums.hip; forced: rsp_debug_set("row_segments", 2)) against
numpy.bincount: random row counts above one block (2..20 blocks, now and then 60+), 1..3000 columns, column lengths
from 0 to hundreds, repeated rows, row indices outside [0, nrow), and every fifth matrix with a column whose rows do
not ascend (the handle must notice and take another form).  Prints one JSON line.
    python3 tools/soak_row_segments.py [cases] [seed]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rcppsparse_amd import capi, synth

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
capi.load()
capi.set_row_segments(2)
forms, worst, t0 = {}, 0.0, time.time()
for case in range(cases):
    rng = np.random.default_rng(seed0 * 100_003 + case)
    nrow = int(rng.integers(16_385, 330_000)) if case % 9 else int(rng.integers(1_000_000, 1_200_000))
    ncol = int(rng.integers(1, 3000))
    mean = int(rng.integers(0, 400)) if case % 4 else int(rng.integers(0, 6))
    counts = rng.poisson(mean, size=ncol).astype(np.int64)
    counts[rng.random(ncol) < 0.05] = 0
    if case % 6 == 0:
        counts[int(rng.integers(0, ncol))] = int(rng.integers(1000, 60_000))        # one long column
    col = np.repeat(np.arange(ncol, dtype=np.int64), counts)
    row = rng.integers(0, nrow, size=col.size, dtype=np.int64)
    if case % 3 == 0 and row.size > 20:
        row[1::5] = row[0::5][:row[1::5].size]
    if case % 7 == 0:
        row[rng.random(row.size) < 0.02] = nrow + int(rng.integers(0, 1000))
        row[rng.random(row.size) < 0.02] = -int(rng.integers(1, 1000))
    order = np.lexsort((row, col))
    col, row = col[order], row[order].astype(np.int32)
    p = np.zeros(ncol + 1, dtype=np.int64)
    np.add.at(p, col + 1, 1)
    p = np.cumsum(p).astype(np.int32)
    nnz = int(p[-1])
    if nnz == 0:
        continue
    unsorted = False
    if case % 5 == 4:
        lens = np.diff(p)
        c = int(np.argmax(lens))
        if lens[c] >= 2 and row[p[c]] != row[p[c + 1] - 1]:
            row[p[c]], row[p[c + 1] - 1] = row[p[c + 1] - 1], row[p[c]]            # largest row first: a descent inside the column
            unsorted = True
    x = synth.gen_values(nnz, seed=case, kind=case % 2)
    h = capi.DeviceCSC(x, p, (nrow, ncol), i=row)
    got, again = h.row_sums(), h.row_sums()
    form = h.row_form()
    h.close()
    forms[form] = forms.get(form, 0) + 1
    assert (form == "segments") != unsorted, (case, form, unsorted)
    assert got.tobytes() == again.tobytes(), case
    keep = (row >= 0) & (row < nrow)
    ref = np.bincount(row[keep], weights=x[keep], minlength=nrow)
    scale = np.bincount(row[keep], weights=np.abs(x[keep]), minlength=nrow)
    err = np.abs(got - ref)
    if np.any(err > 1e-12 * scale):
        r = int(np.flatnonzero(err > 1e-12 * scale)[0])
        print(json.dumps({"FAILED": case, "row": r, "got": float(got[r]), "ref": float(ref[r]), "nrow": nrow, "ncol": ncol, "form": form}))
        sys.exit(1)
    worst = max(worst, float(np.max(err / np.maximum(scale, 1e-300))))
    if case % 20 == 19:
        print(f"# {case + 1} cases, {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
print(json.dumps({"cases": cases, "seed": seed0, "forms": forms, "max_err_over_l1": worst, "seconds": round(time.time() - t0, 1)}))
