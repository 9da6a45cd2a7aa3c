#!/usr/bin/env python3
"""One-off checks of the "next" entry points at the 32-bit limit (nnz = 2^31 - 1), run by hand on one GPU
(about 4 GPU-minutes, ~60 GB of HBM).  Since round 3 the same shapes are pinned inside the -m gpu suite against the
ORACLE (tests/test_gpu_fullsize_next.py); this script keeps the torch-based whole-array comparisons.

  * rowSums in its direct, partition and block-sort forms against torch's index_add_ (1e-11 of the row's 1-norm),
    bit-stable run to run;
  * row-restricted column sums in the three bitmap regimes (L1 / LDS / L2 probes) against the plain column
    sums of the masked values;
  * crossprod, both kernels bit for bit, on 48 columns of 45e6 rows.

Round 2: all pass; the first run of the rowSums part faulted in the block-sort form (a 32-bit entry cursor
wrapped past 2^31 - 1; fixed in csrc/rowsums.hip, the same pattern in csrc/crossprod.hip).

    python3 tools/check_at_int32_limit.py [rows] [masked] [crossprod]
"""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rcppsparse_amd import capi
L = capi.load()
nnz = 2**31 - 1

def check_rows():
    for nrow in (3, 16_384, 10_000_000, 13_631_488, 20_000_000):
        ncol = 1_000_000
        p = np.linspace(0, nnz, ncol + 1).astype(np.int64).astype(np.int32)
        pt = torch.from_numpy(p).cuda()
        xt = torch.empty(nnz, dtype=torch.float64, device="cuda")
        it = torch.empty(nnz, dtype=torch.int32, device="cuda")
        capi.gen_values_device(xt, 5, 0, 0)
        capi.gen_row_indices_device(it, pt, nrow, 5)
        got = capi.row_sums_device(xt, it, nrow)
        again = capi.row_sums_device(xt, it, nrow)
        ref = torch.zeros(nrow, dtype=torch.float64, device="cuda")
        scale = torch.zeros(nrow, dtype=torch.float64, device="cuda")
        step = 200_000_000
        for a in range(0, nnz, step):
            idx = it[a:a + step].to(torch.int64)
            ref.index_add_(0, idx, xt[a:a + step])
            scale.index_add_(0, idx, xt[a:a + step].abs())
            del idx
        err = ((got - ref).abs() / scale.clamp_min(1e-300)).max().item()
        print(json.dumps({"row_sums_nrow": nrow, "nnz": nnz, "bit_stable": bool(torch.equal(got, again)),
                          "max_rel_err_vs_index_add": err}), flush=True)
        del xt, it, got, again, ref, scale
        torch.cuda.empty_cache()

def check_masked():
    # --- row-restricted sums at 2^31-1 entries: three bitmap sizes (L1 / LDS / L2 probes)
    for nrow, ncol in ((3000, 1_000_000), (1_000_000, 1_000_000), (10_000_000, 500_000)):
        p = np.linspace(0, nnz, ncol + 1).astype(np.int64).astype(np.int32)
        pt = torch.from_numpy(p).cuda()
        xt = torch.empty(nnz, dtype=torch.float64, device="cuda"); capi.gen_values_device(xt, 9, 0, 0)
        it = torch.empty(nnz, dtype=torch.int32, device="cuda"); capi.gen_row_indices_device(it, pt, nrow, 9)
        rng = np.random.default_rng(nrow)
        rows = np.flatnonzero(rng.random(nrow) < 0.5)
        bits = capi.row_set_bitmap(rows, nrow)
        bt = torch.from_numpy(bits).cuda()
        res = {}
        for comp in (False, True):
            got = capi.column_sums_in_rows_device(xt, it, pt, nrow, bt, comp)
            # reference: zero the excluded entries with torch, then the plain column sums (checked at this size elsewhere)
            xm = xt.clone()
            step = 250_000_000
            bt64 = bt.to(torch.int64)
            for a in range(0, nnz, step):
                ii = it[a:a + step].to(torch.int64)
                inset = ((bt64[ii >> 5] >> (ii & 31)) & 1).bool()
                keep = ~inset if comp else inset
                xm[a:a + step] *= keep
                del ii, inset, keep
            ref = capi.column_sums_device(xm, pt)
            sc = capi.column_sums_device(xm.abs_(), pt)
            err = ((got - ref).abs() / sc.clamp_min(1e-300)).max().item()
            res["complement" if comp else "in_set"] = err
            del xm, ref, sc, got
        print(json.dumps({"masked_nrow": nrow, "ncol": ncol, "nnz": nnz, **res}), flush=True)
        del xt, it, pt, bt
        torch.cuda.empty_cache()

def check_crossprod():
    # --- crossprod with 2^31-1 entries in 48 columns of 45e6 rows: both kernels, bit for bit
    nrow, ncol = 45_000_000, 48
    p = np.linspace(0, nnz, ncol + 1).astype(np.int64).astype(np.int32)
    pt = torch.from_numpy(p).cuda()
    xt = torch.empty(nnz, dtype=torch.float64, device="cuda"); capi.gen_values_device(xt, 3, 0, 0)
    it = torch.empty(nnz, dtype=torch.int32, device="cuda"); capi.gen_row_indices_device(it, pt, nrow, 3)
    t0 = time.time(); a = capi.crossprod_device(xt, it, pt, nrow); torch.cuda.synchronize(); t1 = time.time()
    b = capi.crossprod_device(xt, it, pt, nrow, tiles=True); torch.cuda.synchronize(); t2 = time.time()
    diag = torch.zeros(ncol, dtype=torch.float64, device="cuda")
    sq = capi.column_sums_device(xt * xt, pt)
    print(json.dumps({"crossprod": f"{nrow}x{ncol}", "nnz": nnz, "rows_s": t1 - t0, "tiles_s": t2 - t1,
                      "same_bits": bool(torch.equal(a, b)),
                      "diag_rel_err_vs_sum_of_squares": ((a.diagonal() - sq).abs() / sq).max().item(),
                      "symmetric": bool(torch.equal(a, a.T))}), flush=True)


if __name__ == "__main__":
    what = sys.argv[1:] or ["rows", "masked", "crossprod"]
    if "rows" in what: check_rows()
    if "masked" in what: check_masked()
    if "crossprod" in what: check_crossprod()
