"""GPU parity of Matrix::crossprod (reference inst/include/RcppSparse.h:159-194; "next" row f3)
against the oracle's pairwise sorted-merge loop.  Both device kernels (the row-major path that
needs a workspace, and the scratch-free tile kernel) accumulate the products of the common rows
in ascending row order with a separate multiply and add -- the reference's order -- so finite
results must be bit-identical, not merely within tolerance."""
import numpy as np
import pytest
import scipy.sparse as sp

import oracle
from rcppsparse_amd import capi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked tests need a GPU")
    capi.load()
    return torch


@pytest.mark.parametrize("nrow,ncol,density", [
    (10, 10, 0.1), (5, 5, 0.5), (40, 30, 0.15), (1000, 64, 0.05), (1000, 65, 0.05), (300, 200, 0.3),
    (20_000, 130, 0.01), (64, 257, 0.9), (100_000, 8, 0.2), (3, 500, 0.6), (2_000_000, 70, 0.0003),
    (500, 150, 1.0), (7, 1, 0.5), (1, 90, 0.7), (4000, 1100, 0.02),
])
@pytest.mark.parametrize("tiles", [False, True], ids=["rows", "tiles"])
def test_crossprod_bit_exact_vs_oracle(torch_cuda, nrow, ncol, density, tiles):
    torch = torch_cuda
    m = synth.rsparsematrix(nrow, ncol, density=density, seed=nrow % 89 + ncol, kind=0)
    x, i, p = m["x"], m["i"], m["p"]
    ref = oracle.crossprod(x, i, p)
    xt, it, pt = torch.from_numpy(x).cuda(), torch.from_numpy(i).cuda(), torch.from_numpy(p).cuda()
    if x.size == 0:
        xt = torch.zeros(2, dtype=torch.float64, device="cuda")[:0]
        it = torch.zeros(2, dtype=torch.int32, device="cuda")[:0]
    # (a shape with few long columns would go to the tall MFMA form, which is tested further down: this test
    # is about the two bit-identical kernels)
    capi.set_crossprod_exact(True)
    try:
        got = capi.crossprod_device(xt, it, pt, nrow, tiles=tiles).cpu().numpy().T   # column-major -> row-major view
    finally:
        capi.set_crossprod_exact(False)
    assert got.shape == (ncol, ncol)
    assert np.array_equal(got, ref), float(np.max(np.abs(got - ref)))
    assert np.array_equal(got, got.T)                              # mirrored exactly
    A = sp.csc_matrix((x, i, p), shape=(nrow, ncol))
    dense = (A.T @ A).toarray()
    assert np.allclose(got, dense, rtol=1e-12, atol=1e-12)
    if not tiles:
        capi.set_crossprod_exact(True)
        try:
            h = capi.DeviceCSC(x, p, (nrow, ncol), i=i)
            via_handle = h.crossprod()
            h.close()
        finally:
            capi.set_crossprod_exact(False)
        assert np.array_equal(via_handle, ref)


@pytest.mark.parametrize("tiles", [False, True], ids=["rows", "tiles"])
def test_crossprod_nonfinite_entries_only_meet_stored_entries(torch_cuda, tiles):
    torch = torch_cuda
    m = synth.rsparsematrix(200, 70, density=0.1, seed=9)
    x, i, p = m["x"].copy(), m["i"], m["p"]
    x[5] = np.inf
    x[40] = np.nan
    ref = oracle.crossprod(x, i, p)
    got = capi.crossprod_device(torch.from_numpy(x).cuda(), torch.from_numpy(i).cuda(),
                                torch.from_numpy(p).cuda(), 200, tiles=tiles).cpu().numpy().T
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    ok = ~np.isnan(ref)
    assert np.array_equal(got[ok], ref[ok])


@pytest.mark.parametrize("tiles", [False, True], ids=["rows", "tiles"])
def test_crossprod_subnormal_and_huge_products_round_like_the_reference(torch_cuda, tiles):
    # products and partial sums in the subnormal range, cancellation, and values near overflow:
    # the accumulation must round exactly like the reference's `res += x1 * x2` in double
    torch = torch_cuda
    nrow, ncol = 400, 40
    m = synth.rsparsematrix(nrow, ncol, density=0.4, seed=77)
    x, i, p = m["x"].copy(), m["i"], m["p"]
    col_scale = np.array([1e-160, 3e-158, 1e-155, 1.0, 1e150, -1e-157, 7e153, 1e-150])[np.arange(ncol) % 8]
    x = (x + 0.37) * np.repeat(col_scale, np.diff(p))
    with np.errstate(over="ignore", invalid="ignore"):
        ref = oracle.crossprod(x, i, p)
    assert np.any((np.abs(ref) > 0) & (np.abs(ref) < 2.3e-308))     # subnormal results are present
    got = capi.crossprod_device(torch.from_numpy(x).cuda(), torch.from_numpy(i).cuda(),
                                torch.from_numpy(p).cuda(), nrow, tiles=tiles).cpu().numpy().T
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    ok = ~np.isnan(ref)
    assert np.array_equal(got[ok], ref[ok])
    assert np.array_equal(np.signbit(got[ok]), np.signbit(ref[ok]))


def test_crossprod_wide_matrix_splits_result_columns(torch_cuda):
    # more columns than one wave's LDS accumulators hold (8192): the result column is split
    torch = torch_cuda
    nrow, ncol = 300, 9000
    m = synth.rsparsematrix(nrow, ncol, density=0.004, seed=5)
    x, i, p = m["x"], m["i"], m["p"]
    got = capi.crossprod_device(torch.from_numpy(x).cuda(), torch.from_numpy(i).cuda(),
                                torch.from_numpy(p).cuda(), nrow).cpu().numpy()
    A = sp.csc_matrix((x, i, p), shape=(nrow, ncol))
    dense = (A.T @ A).toarray()
    assert np.array_equal(got, got.T)
    assert np.allclose(got, dense, rtol=1e-12, atol=1e-12)
    sub = 40
    ref = oracle.crossprod(x[:p[sub]], i[:p[sub]], np.ascontiguousarray(p[:sub + 1]))
    assert np.array_equal(got[:sub, :sub], ref)


def test_crossprod_full_lds_width(torch_cuda):
    # exactly 8192 columns: one slice per result column, 64 KB of LDS accumulators per wave
    torch = torch_cuda
    nrow, ncol = 60, 8192
    m = synth.rsparsematrix(nrow, ncol, density=0.01, seed=11)
    x, i, p = m["x"], m["i"], m["p"]
    got = capi.crossprod_device(torch.from_numpy(x).cuda(), torch.from_numpy(i).cuda(),
                                torch.from_numpy(p).cuda(), nrow)
    assert bool(torch.equal(got, got.T))
    A = sp.csc_matrix((x, i, p), shape=(nrow, ncol))
    for lo in (0, 4000, 8100):
        hi = min(ncol, lo + 92)
        want = (A.T @ A[:, lo:hi]).toarray()              # rows: all columns, cols: lo..hi
        assert np.allclose(got[lo:hi, :].cpu().numpy().T, want, rtol=1e-12, atol=1e-12)


def test_crossprod_workspace_too_small_is_reported(torch_cuda):
    torch = torch_cuda
    m = synth.rsparsematrix(100, 20, density=0.2, seed=2)
    xt, it, pt = (torch.from_numpy(m[k]).cuda() for k in ("x", "i", "p"))
    ws = torch.empty(16, dtype=torch.uint8, device="cuda")
    with pytest.raises(capi.RspError, match="workspace too small"):
        capi.crossprod_device(xt, it, pt, 100, workspace=ws)


def test_crossprod_needs_row_indices(torch_cuda):
    m = synth.rsparsematrix(50, 10, density=0.2, seed=1)
    h = capi.DeviceCSC(m["x"], m["p"], m["Dim"])
    with pytest.raises(capi.RspError):
        h.crossprod()
    h.close()


# ---- the tall form (ncol <= 512, columns of >= 4096 entries): matrix cores, tolerance instead of bits ----

TALL_SHAPES = [(41_000, 100, 0.1), (50_000, 7, 0.1), (60_000, 129, 0.1), (50_000, 192, 0.1), (45_000, 200, 0.1),
               (42_000, 256, 0.1), (400_000, 1, 0.5), (300_000, 16, 0.2), (300_000, 17, 0.15), (250_000, 48, 0.2), (200_000, 64, 0.25),
               (200_000, 65, 0.2), (150_000, 100, 0.3), (150_000, 128, 0.25), (3_000_000, 20, 0.02),
               (250_000, 256, 0.0166),   # (7813 panels of 32 rows: the last workgroup's range is ONE panel)
               (45_000, 300, 0.1), (41_000, 512, 0.1), (43_000, 385, 0.1)]   # (24 / 32 tiles: panels of 16 rows, three / four workgroups per range)


@pytest.mark.parametrize("nrow,ncol,density", TALL_SHAPES)
def test_crossprod_tall_form_within_tolerance_and_deterministic(torch_cuda, nrow, ncol, density):
    """Few long columns: the workspace form densifies 64 rows at a time and runs t(P) P as f64 MFMA, every
    workgroup over its own rows, results added in workgroup order.  Not the reference's order, so the bar is
    the floating-point one: 1e-12 of sum|x1 x2| per entry against the oracle's merges (signed data), the
    same bits on every run, an exactly symmetric result; and with rsp_set_crossprod_exact(1) the
    bit-identical form is back."""
    torch = torch_cuda
    m = synth.rsparsematrix(nrow, ncol, density=density, seed=nrow % 97 + ncol, kind=0)
    x, i, p = m["x"], m["i"], m["p"]
    assert x.size // ncol >= 4096 and capi.crossprod_form(nrow, ncol, x.size) == "tall"   # (the shape does select the tall form)
    ref = oracle.crossprod(x, i, p)
    scale = oracle.crossprod(np.abs(x), i, p)
    xt, it, pt = torch.from_numpy(x).cuda(), torch.from_numpy(i).cuda(), torch.from_numpy(p).cuda()
    got = capi.crossprod_device(xt, it, pt, nrow).cpu().numpy().T
    again = capi.crossprod_device(xt, it, pt, nrow).cpu().numpy().T
    assert got.tobytes() == again.tobytes()
    assert np.array_equal(got, got.T)
    assert np.all(np.abs(got - ref) <= 1e-12 * scale), float(np.max(np.abs(got - ref) / np.maximum(scale, 1e-300)))
    assert np.all(got[scale == 0] == 0)                              # column pairs without a common row
    h = capi.DeviceCSC(x, p, (nrow, ncol), i=i)
    via_handle = h.crossprod()
    h.close()
    assert via_handle.tobytes() == got.tobytes()                     # same form behind the handle
    capi.set_crossprod_exact(True)
    try:
        exact = capi.crossprod_device(xt, it, pt, nrow).cpu().numpy().T
    finally:
        capi.set_crossprod_exact(False)
    assert np.array_equal(exact, ref)


def test_crossprod_tall_form_steps_aside_for_nonfinite_values(torch_cuda):
    """A structural zero times an infinity would be NaN where the reference has no product at all: with any
    NaN / Inf in x the tall kernels exit and the bit-identical kernel (standing by on the same row-major
    form) produces the result -- the reference's, bit for bit."""
    torch = torch_cuda
    nrow, ncol = 250_000, 40
    m = synth.rsparsematrix(nrow, ncol, density=0.2, seed=5, kind=0)
    x, i, p = m["x"].copy(), m["i"], m["p"]
    x[1234] = np.inf
    x[x.size // 2] = np.nan
    x[-7] = -np.inf
    ref = oracle.crossprod(x, i, p)
    xt, it, pt = torch.from_numpy(x).cuda(), torch.from_numpy(i).cuda(), torch.from_numpy(p).cuda()
    got = capi.crossprod_device(xt, it, pt, nrow).cpu().numpy().T
    assert np.array_equal(got, ref, equal_nan=True)
    h = capi.DeviceCSC(x, p, (nrow, ncol), i=i)                      # (behind a handle: the host looks at the tall form's flag)
    via_handle = h.crossprod()
    h.close()
    assert np.array_equal(via_handle, ref, equal_nan=True)
    finite = np.isfinite(ref)
    assert finite.sum() > 0.8 * ref.size                              # the non-finite values touched only their own columns


def test_crossprod_tall_form_rows_without_entries_and_ragged_last_panel(torch_cuda):
    """Row panels (64 rows) with no entry at all are skipped, the last panel is partial, and all entries may
    sit in a few rows."""
    torch = torch_cuda
    nrow, ncol = 1_000_003, 24
    rng = np.random.default_rng(3)
    cols = []
    for c in range(ncol):
        lo = int(rng.integers(0, nrow - 200_000))
        rows = np.sort(rng.choice(np.arange(lo, lo + 200_000), size=40_000, replace=False))
        rows[-1] = nrow - 1 if c % 3 == 0 else rows[-1]
        cols.append(np.unique(rows))
    i = np.concatenate(cols).astype(np.int32)
    p = np.concatenate(([0], np.cumsum([len(c) for c in cols]))).astype(np.int32)
    x = rng.standard_normal(i.size)
    ref = oracle.crossprod(x, i, p)
    scale = oracle.crossprod(np.abs(x), i, p)
    got = capi.crossprod_device(torch.from_numpy(x).cuda(), torch.from_numpy(i).cuda(), torch.from_numpy(p).cuda(),
                                nrow).cpu().numpy().T
    assert np.all(np.abs(got - ref) <= 1e-12 * scale)
    assert np.all(got[scale == 0] == 0)


# ---- 97-512 columns: the tall form finds its 32-row (from 257 columns on: 16-row) panels through a panel table ----

@pytest.mark.parametrize("ncol", [200, 256, 176, 120, 400])
def test_crossprod_panel_table_form_gaps_ragged_end_and_round3_kernel_agree(torch_cuda, ncol):
    """Panels without entries never enter the pipeline (has[]), whether the gap is two panels or 3 400 and wherever a
    workgroup's range begins; the last panel is partial.  Against the oracle within the tall form's tolerance, the
    same bits on every run, and the kernel it replaces (RSP_CROSSPROD_PANEL_TABLE=0) within the same tolerance."""
    import os
    torch = torch_cuda
    nrow = 200_003
    bands = [(0, 9_000), (9_100, 9_164), (60_000, 90_000), (199_000, nrow - 1)]
    rng = np.random.default_rng(11 + ncol)
    cols = []
    for c in range(ncol):
        parts = [np.sort(rng.choice(np.arange(lo, hi), size=n, replace=False))
                 for (lo, hi), n in zip(bands, (2500, 20, 1500, 300))]
        rows = np.concatenate(parts)
        if c % 5 == 0:
            rows = np.append(rows, nrow - 1)
        cols.append(rows)
    i = np.concatenate(cols).astype(np.int32)
    p = np.concatenate(([0], np.cumsum([len(c) for c in cols]))).astype(np.int32)
    x = rng.standard_normal(i.size)
    if ncol > 256:
        os.environ["RSP_CROSSPROD_TALL_ALWAYS"] = "1"     # (400 columns of 4320 entries: the cost model would take the exact form)
    try:
        assert capi.crossprod_form(nrow, ncol, x.size) == "tall"
        ref = oracle.crossprod(x, i, p)
        scale = oracle.crossprod(np.abs(x), i, p)
        xt, it, pt = torch.from_numpy(x).cuda(), torch.from_numpy(i).cuda(), torch.from_numpy(p).cuda()
        got = capi.crossprod_device(xt, it, pt, nrow).cpu().numpy().T
        again = capi.crossprod_device(xt, it, pt, nrow).cpu().numpy().T
        assert got.tobytes() == again.tobytes()
        assert np.array_equal(got, got.T)
        assert np.all(np.abs(got - ref) <= 1e-12 * scale), float(np.max(np.abs(got - ref) / np.maximum(scale, 1e-300)))
        os.environ["RSP_CROSSPROD_PANEL_TABLE"] = "0"                # (above 256 columns: the exact form)
        try:
            old = capi.crossprod_device(xt, it, pt, nrow).cpu().numpy().T
        finally:
            del os.environ["RSP_CROSSPROD_PANEL_TABLE"]
        assert np.all(np.abs(old - ref) <= 1e-12 * scale)
    finally:
        os.environ.pop("RSP_CROSSPROD_TALL_ALWAYS", None)


@pytest.mark.parametrize("ncol", [16 * rt - 5 for rt in range(17, 33)] + [273, 497])   # (every real tile count of the two layouts; 273 / 497: a last tile of ONE column)
def test_crossprod_wide_forms_at_every_real_tile_count(torch_cuda, ncol):
    """Round 5: the 24 / 32-tile kernels are instantiated per REAL tile count -- pairs per tile row = tiles / 2 + 1, tile rows
    meeting modulo the real count, the workgroups of a panel range sharing the real tile rows: every count from 17 to 32,
    even and odd, and a last tile of a single column.  Against SciPy's product within the tall form's tolerance, the same bits
    twice, symmetric."""
    torch = torch_cuda
    nrow = 43_000
    m = synth.rsparsematrix(nrow, ncol, density=0.1, seed=ncol, kind=0)
    x, i, p = m["x"], m["i"], m["p"]
    assert capi.crossprod_form(nrow, ncol, x.size) == "tall"
    # (the reference here is SciPy's sparse product, not the oracle's merges -- 400 columns squared times 4300 entries are
    # seconds per case in the oracle; the bar is the tall form's tolerance, to which any summation order agrees, and the
    # oracle's crossprod is itself held against SciPy in tests/test_oracle.py::test_crossprod_restatement_against_scipy)
    import scipy.sparse as sp
    A = sp.csc_matrix((x, i, p), shape=(nrow, ncol))
    B = sp.csc_matrix((np.abs(x), i, p), shape=(nrow, ncol))
    ref = np.asarray((A.T @ A).todense())
    scale = np.asarray((B.T @ B).todense())
    xt, it, pt = torch.from_numpy(x).cuda(), torch.from_numpy(i).cuda(), torch.from_numpy(p).cuda()
    got = capi.crossprod_device(xt, it, pt, nrow).cpu().numpy().T
    again = capi.crossprod_device(xt, it, pt, nrow).cpu().numpy().T
    assert got.tobytes() == again.tobytes()
    assert np.array_equal(got, got.T)
    assert np.all(np.abs(got - ref) <= 1e-12 * scale), float(np.max(np.abs(got - ref) / np.maximum(scale, 1e-300)))
    assert np.all(got[scale == 0] == 0)


@pytest.mark.parametrize("ncol,order", [
    (257, "tail-first"),      # 24 tiles, three workgroups per panel range: part 2 densifies tiles 16..23 and 0..11
    (512, "head-late"),       # 32 tiles, four workgroups: part 1 densifies tiles 8..31 -- column 0 is not among them
    (300, "tail-first"), (400, "head-late")])
def test_crossprod_wide_forms_columns_in_disjoint_row_windows(torch_cuda, ncol, order):
    """Round 5 (found by tools/soak_crossprod_tall.py, 257 columns): at 24 / 32 tiles a workgroup densifies only the tiles its
    tile rows meet, and a panel that held entries of OTHER tiles only entered its pipeline without leaving an entry for
    the lanes that have none to read -- they went on reading entry 0 of the matrix, and when that entry's row lay in the panel
    being filled, x[0] stood in every column of that row: x[0]^2 in column pairs that share no row.  Every column dense in a
    row window of its own, the windows in an order that puts column 0's first row behind a long run of panels in which a
    workgroup finds nothing of its own tiles: every off-diagonal element is exactly zero."""
    import os
    torch = torch_cuda
    length = 4100
    if order == "tail-first":          # columns 192.. first (tiles 12..15: not densified by part 2 of 24 tiles), then 0, 1, ...
        seq = list(range(192, min(ncol, 256))) + list(range(0, 192)) + list(range(256, ncol))
    else:                              # columns 1..127 first (tiles 0..7: not densified by part 1 of 32 tiles), then 0, then the rest
        seq = list(range(1, 128)) + [0] + list(range(128, ncol))
    assert sorted(seq) == list(range(ncol))
    start = np.empty(ncol, dtype=np.int64)
    start[np.array(seq)] = np.arange(ncol, dtype=np.int64) * length + 5        # (windows do not start on a panel edge)
    nrow = ncol * length + 40
    i = (start[:, None] + np.arange(length)[None, :]).reshape(-1).astype(np.int32)
    p = (np.arange(ncol + 1, dtype=np.int64) * length).astype(np.int32)
    x = np.random.default_rng(ncol).standard_normal(i.size) * 300.0
    os.environ["RSP_CROSSPROD_TALL_ALWAYS"] = "1"     # (no two columns share a row: the cost model would take the exact form)
    try:
        assert capi.crossprod_form(nrow, ncol, x.size) == "tall"
        xt, it, pt = torch.from_numpy(x).cuda(), torch.from_numpy(i).cuda(), torch.from_numpy(p).cuda()
        got = capi.crossprod_device(xt, it, pt, nrow).cpu().numpy()
    finally:
        os.environ.pop("RSP_CROSSPROD_TALL_ALWAYS", None)
    diag = np.array([np.dot(x[p[c]:p[c + 1]], x[p[c]:p[c + 1]]) for c in range(ncol)])
    off = got - np.diag(np.diag(got))
    assert not off.any(), (int(np.count_nonzero(off)), float(np.abs(off).max()), np.argwhere(off)[:4].tolist())
    assert np.all(np.abs(np.diag(got) - diag) <= 1e-12 * diag)


@pytest.mark.parametrize("ncol", [256, 180, 112, 330])
def test_crossprod_panel_table_form_steps_aside_for_nonfinite_values(torch_cuda, ncol):
    """16 / 12 / 8 column tiles: the panel-table kernel looks at its sums, not at every value: a NaN made by a structural zero
    meeting an infinity stays a NaN, the flag goes up and the bit-identical kernels produce the reference's result.
    Also with the non-finite value in the last entry of the last column, and with a value whose products overflow."""
    torch = torch_cuda
    nrow = 45_000
    m = synth.rsparsematrix(nrow, ncol, density=0.1, seed=9, kind=0)
    x0, i, p = m["x"], m["i"], m["p"]
    assert capi.crossprod_form(nrow, ncol, x0.size) == "tall"
    it, pt = torch.from_numpy(i).cuda(), torch.from_numpy(p).cuda()
    for where, value in ((1234, np.inf), (x0.size - 1, np.nan), (x0.size // 3, -np.inf), (777, 1e200)):
        x = x0.copy()
        x[where] = value
        if value == 1e200:
            x[where + 1] = 1e200                                          # (its square is an infinity only in the sum)
        with np.errstate(over="ignore", invalid="ignore"):
            ref = oracle.crossprod(x, i, p)
        got = capi.crossprod_device(torch.from_numpy(x).cuda(), it, pt, nrow).cpu().numpy().T
        assert np.array_equal(got, ref, equal_nan=True), (where, value)
        if value != 1e200:
            continue
        # behind a handle the tall form runs alone and the host looks at its flag (rsp_csc_crossprod)
        h = capi.DeviceCSC(x, p, (nrow, ncol), i=i)
        via_handle = h.crossprod()
        h.close()
        assert np.array_equal(via_handle, ref, equal_nan=True)


@pytest.mark.parametrize("ncol", [256, 180, 112, 500, 290, 420])   # (290, 420: the 24 / 32-tile kernels at a partial width)
def test_crossprod_panel_table_form_is_memory_safe_on_invalid_matrices(torch_cuda, ncol):
    """Not a dgCMatrix -- rows that do not ascend, rows outside the matrix, column offsets that go backwards or
    beyond nnz: the result means nothing, but the call returns, reads and writes nothing out of bounds (the entries
    it cannot place are dropped) and the library goes on working."""
    torch = torch_cuda
    nrow = 45_000
    m = synth.rsparsematrix(nrow, ncol, density=0.1, seed=10, kind=0)
    x, i0, p0 = m["x"], m["i"], m["p"]
    rng = np.random.default_rng(0)
    xt = torch.from_numpy(x).cuda()
    cases = []
    i = i0.copy(); rng.shuffle(i[: i.size // 2]); cases.append((i, p0))                      # rows in any order
    i = i0.copy(); i[::7] = nrow + 5; i[3::11] = -3; cases.append((i, p0))                 # rows outside the matrix
    i = i0.copy(); i[:] = 17; cases.append((i, p0))                                         # one row, stored again and again
    p = p0.copy(); p[5] = p[9]; p[100] = x.size + 1000; p[ncol - 3] = -4; cases.append((i0, p))  # offsets that are no offsets
    for i, p in cases:
        out = capi.crossprod_device(xt, torch.from_numpy(i).cuda(), torch.from_numpy(p).cuda(), nrow)
        torch.cuda.synchronize()
        assert out.shape == (ncol, ncol)
    ref = oracle.crossprod(x, i0, p0)
    scale = oracle.crossprod(np.abs(x), i0, p0)
    got = capi.crossprod_device(xt, torch.from_numpy(i0).cuda(), torch.from_numpy(p0).cuda(), nrow).cpu().numpy().T
    assert np.all(np.abs(got - ref) <= 1e-12 * scale)


@pytest.mark.parametrize("ncol", [200, 120, 40, 300])
def test_crossprod_device_entry_is_graph_capture_safe(torch_cuda, ncol):
    """rsp_crossprod_device with a workspace allocates nothing and never synchronises (the exact kernels stand by on
    a device-side flag): capturable into a HIP graph in the panel-table form (16 / 8 tiles) and in the form that walks
    the CSC arrays (3 tiles); a replay reproduces the bits, also with new values in the same buffers (x doubled: every
    sum exactly four times as large)."""
    torch = torch_cuda
    nrow = 45_000
    m = synth.rsparsematrix(nrow, ncol, density=0.1, seed=12, kind=0)
    x, i, p = m["x"], m["i"], m["p"]
    assert capi.crossprod_form(nrow, ncol, x.size) == "tall"
    xt, it, pt = torch.from_numpy(x).cuda(), torch.from_numpy(i).cuda(), torch.from_numpy(p).cuda()
    nbytes = int(capi.load().rsp_crossprod_workspace_bytes(nrow, ncol, x.size))
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    out = torch.zeros((ncol, ncol), dtype=torch.float64, device="cuda")
    eager = capi.crossprod_device(xt, it, pt, nrow, out.clone(), workspace=ws).clone()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        capi.crossprod_device(xt, it, pt, nrow, out, workspace=ws)      # enqueued on the capturing stream
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager)
    xt.mul_(2.0)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager * 4.0)
