"""GPU parity of the row-wise "next" entries: Matrix::rowSums / rowMeans
(reference inst/include/RcppSparse.h:138-156) through rsp_row_sums_device / rsp_csc_row_sums,
against the oracle's scatter loop on the same inputs.  Tolerance: 1e-12 of the row's 1-norm;
bit-identical run to run (no float atomics in global memory: a stable sort by row block + in-order LDS
accumulation for the one-shot entry, a stable sort by row + the column-sum kernels behind the handle)."""
import numpy as np
import pytest

import oracle
from rcppsparse_amd import capi, synth

pytestmark = pytest.mark.gpu
RTOL = 1e-12


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked tests need a GPU")
    capi.load()
    return torch


def row_l1(x, i, nrow):
    return np.bincount(i, weights=np.abs(x), minlength=nrow)


def check(got, x, i, p, nrow):
    ref = oracle.row_sums(x, i, p, nrow)
    scale = row_l1(x, i, nrow)
    assert got.shape == (nrow,)
    assert np.all(np.abs(got - ref) <= RTOL * scale), float(np.max(np.abs(got - ref) / np.maximum(scale, 1e-300)))
    empty = np.bincount(i, minlength=nrow) == 0
    assert np.all(got[empty] == 0.0) and not np.any(np.signbit(got[empty]))


@pytest.mark.parametrize("nrow,ncol,density,kind", [
    (10, 10, 0.1, 0), (40, 30, 0.15, 0), (1000, 200, 0.05, 1), (5000, 3000, 0.002, 0),
    (200_000, 300, 0.01, 0), (64, 20_000, 0.2, 1), (1, 500, 0.7, 0), (100_000, 4, 0.5, 1),
])
def test_row_sums_and_means_match_oracle(torch_cuda, nrow, ncol, density, kind):
    torch = torch_cuda
    m = synth.rsparsematrix(nrow, ncol, density=density, seed=nrow + ncol, kind=kind)
    x, i, p = m["x"], m["i"], m["p"]
    xt, it = torch.from_numpy(x).cuda(), torch.from_numpy(i).cuda()
    got = capi.row_sums_device(xt, it, nrow).cpu().numpy()
    check(got, x, i, p, nrow)
    again = capi.row_sums_device(xt, it, nrow).cpu().numpy()
    assert got.tobytes() == again.tobytes()                       # deterministic
    means = capi.row_sums_device(xt, it, nrow, ncol_for_means=ncol).cpu().numpy()
    assert means.tobytes() == (got / ncol).tobytes()              # RcppSparse.h:153-154 divides the sums
    h = capi.DeviceCSC(x, p, (nrow, ncol), i=i)
    hs, hm = h.row_sums(), h.row_means()
    hs2 = h.row_sums()                                            # second call reuses the cached row-major form
    cs = h.column_sums()                                          # and the column path still works
    h.close()
    # the handle keeps the row-major form (full sort by row), the one-shot entry groups by row block:
    # two deterministic orders, each within tolerance of the oracle, each bit-stable
    check(hs, x, i, p, nrow)
    assert hs.tobytes() == hs2.tobytes() and hm.tobytes() == (hs / ncol).tobytes()
    assert np.allclose(cs, oracle.column_sums(x, p), rtol=0, atol=1e-9)


def test_handle_without_row_indices_fails_loudly(torch_cuda):
    m = synth.rsparsematrix(50, 40, density=0.2, seed=5)
    h = capi.DeviceCSC(m["x"], m["p"], (50, 40))          # no i[]
    with pytest.raises(capi.RspError) as e:
        h.row_sums()
    assert e.value.code == capi.RSP_ERR_BAD_ARG
    h.close()


def test_empty_matrix_rows(torch_cuda):
    h = capi.DeviceCSC(np.array([], dtype=np.float64), np.zeros(8, dtype=np.int32), (9, 7),
                       i=np.array([], dtype=np.int32))
    rs = h.row_sums()
    h.close()
    assert rs.shape == (9,) and np.all(rs == 0.0)


def test_device_row_index_generator_matches_oracle_and_is_valid_csc(torch_cuda):
    torch = torch_cuda
    counts = synth.uniform_counts(3000, 200_000, seed=4, nrow=50_000)
    p = synth.offsets_from_counts(counts)
    it = torch.empty(int(p[-1]), dtype=torch.int32, device="cuda")
    capi.gen_row_indices_device(it, torch.from_numpy(p).cuda(), 50_000, seed=9)
    got = it.cpu().numpy()
    assert got.tobytes() == oracle.gen_row_indices(p, 50_000, 9).tobytes()
    assert got.min() >= 0 and got.max() < 50_000
    for c in (0, 17, 2999):
        seg = got[p[c]:p[c + 1]]
        assert np.all(np.diff(seg) > 0)


def test_row_sums_1e8_against_oracle(torch_cuda):
    """1e6 x 1e5, nnz 1e8 (a tenth of C3, same density per column): whole result vs the oracle."""
    torch = torch_cuda
    nrow, ncol, nnz = 1_000_000, 100_000, 100_000_000
    p = synth.offsets_from_counts(synth.uniform_counts(ncol, nnz, seed=42, nrow=nrow))
    pt = torch.from_numpy(p).cuda()
    xt = torch.empty(nnz, dtype=torch.float64, device="cuda")
    it = torch.empty(nnz, dtype=torch.int32, device="cuda")
    capi.gen_values_device(xt, 42, 0, 0)
    capi.gen_row_indices_device(it, pt, nrow, 42)
    got = capi.row_sums_device(xt, it, nrow).cpu().numpy()
    x = oracle.gen_values(nnz, 42, 0, 0)
    i = oracle.gen_row_indices(p, nrow, 42)
    check(got, x, i, p, nrow)


@pytest.mark.parametrize("nrow,nnz", [(13_700_000, 3_000_000), (40_000_000, 5_000_000), (16_384, 2_000_000),
                                      (16_385, 400_000), (3, 5_000), (49_153, 900_000), (65_536, 900_000), (65_537, 900_000),
                                      (13_631_488, 2_500_000), (27_262_976, 2_500_000), (27_262_977, 2_500_000),
                                      (60_000_000, 3_000_000), (109_051_904, 2_000_000), (109_051_905, 2_000_000),
                                      (150_000_000, 4_000_000)])
def test_row_sums_forms_by_row_count(torch_cuda, nrow, nnz):
    """The one-shot entry's forms by row count: up to 4 blocks of 16384 rows nothing is regrouped (every block's
    workgroups scan x / i as they are); up to 832 blocks (1.36e7 rows) the hand-written tile partition regroups by
    block; up to 8 x 832 blocks (1.09e8 rows) it regroups by coarse block of 2 / 4 / 8 row blocks and every row
    block picks its entries out of its coarse block's; above that two partition passes (buckets of 512 blocks,
    then blocks).  Shapes on both sides of every edge.  Same oracle, same tolerance, bit-stable.  All hand-written:
    no library sort anywhere on the row-wise path."""
    torch = torch_cuda
    ncol = 2_000
    p = synth.offsets_from_counts(synth.uniform_counts(ncol, nnz, seed=nrow % 1000, nrow=nrow))
    pt = torch.from_numpy(p).cuda()
    xt = torch.empty(nnz, dtype=torch.float64, device="cuda")
    it = torch.empty(nnz, dtype=torch.int32, device="cuda")
    capi.gen_values_device(xt, 7, 0, 0)
    capi.gen_row_indices_device(it, pt, nrow, 7)
    got = capi.row_sums_device(xt, it, nrow).cpu().numpy()
    again = capi.row_sums_device(xt, it, nrow).cpu().numpy()
    assert got.tobytes() == again.tobytes()
    x = oracle.gen_values(nnz, 7, 0, 0)
    i = oracle.gen_row_indices(p, nrow, 7)
    check(got, x, i, p, nrow)
    # entries whose row index is not in [0, nrow) -- not a valid dgCMatrix -- are left out, not added elsewhere
    # (in every flavour: just past the end, negative, and with upper bits that would sort them among the valid ones)
    it2 = it.clone()
    it2[::1000] = nrow + 5
    it2[1::1000] = -3
    it2[2::1000] = torch.arange(0, nnz, 1000, device="cuda", dtype=torch.int64)[: it2[2::1000].numel()].to(torch.int32) * 37 + 2 * nrow
    it2[3::1000] = -(torch.arange(0, nnz, 1000, device="cuda", dtype=torch.int64)[: it2[3::1000].numel()].to(torch.int32) * 53) - 1
    got2 = capi.row_sums_device(xt, it2, nrow).cpu().numpy()
    keep = np.ones(nnz, dtype=bool)
    for k in range(4):
        keep[k::1000] = False
    ref2 = np.bincount(i[keep], weights=x[keep], minlength=nrow)
    scale2 = np.bincount(i[keep], weights=np.abs(x[keep]), minlength=nrow)
    assert np.all(np.abs(got2 - ref2) <= 1e-11 * np.maximum(scale2, 1e-300) + 1e-300)


def test_handle_row_sums_leave_out_invalid_row_indices(torch_cuda):
    """The handle's row-major form sorts by row; entries whose row index is not in [0, nrow) must end up
    behind every row's range whatever their upper bits are (they get the key nrow), not among the rows."""
    rng = np.random.default_rng(77)
    nrow, ncol, nnz = 70_001, 300, 150_000
    x = rng.standard_normal(nnz)
    i = rng.integers(0, nrow, nnz).astype(np.int32)
    bad = rng.random(nnz) < 0.1
    i[bad] = rng.choice([nrow, nrow + 1, 2 * nrow + 3, 2**30 + 17, -1, -nrow, -2**31], size=int(bad.sum())).astype(np.int32)
    p = np.concatenate(([0], np.sort(rng.integers(0, nnz + 1, ncol - 1)), [nnz])).astype(np.int32)
    h = capi.DeviceCSC(x, p, (nrow, ncol), i=i)
    hs, hs2 = h.row_sums(), h.row_sums()
    hm = h.row_means()
    cs = h.column_sums()
    h.close()
    ref = np.bincount(i[~bad], weights=x[~bad], minlength=nrow)
    scale = np.bincount(i[~bad], weights=np.abs(x[~bad]), minlength=nrow)
    assert hs.tobytes() == hs2.tobytes() and hm.tobytes() == (hs / ncol).tobytes()
    assert np.all(np.abs(hs - ref) <= RTOL * scale), float(np.max(np.abs(hs - ref)))
    assert np.allclose(cs, oracle.column_sums(x, p), rtol=0, atol=1e-9)    # column sums do not look at i[]


def test_row_sums_with_arrays_that_are_only_4_and_8_byte_aligned(torch_cuda):
    """x and i as an R session hands them over need not be 16-byte aligned: views that start one element into a
    buffer go through the forms that load 16 bytes per lane (the histogram pass) and the others alike."""
    torch = torch_cuda
    rng = np.random.default_rng(12)
    for nrow, nnz in ((1000, 200_001), (300_000, 700_003), (14_000_000, 300_001)):
        x = rng.standard_normal(nnz + 3)
        i = rng.integers(0, nrow, nnz + 3).astype(np.int32)
        xt, it = torch.from_numpy(x).cuda(), torch.from_numpy(i).cuda()
        for off in (1, 3):
            got = capi.row_sums_device(xt[off:off + nnz], it[off:off + nnz], nrow).cpu().numpy()
            ref = np.bincount(i[off:off + nnz], weights=x[off:off + nnz], minlength=nrow)
            scale = np.bincount(i[off:off + nnz], weights=np.abs(x[off:off + nnz]), minlength=nrow)
            assert np.all(np.abs(got - ref) <= RTOL * scale), (nrow, off)


@pytest.mark.parametrize("pattern", ["uniform", "one_row", "sorted", "edge_blocks", "one_block_dense", "all_invalid",
                                     "first_tiles_invalid", "mostly_invalid"])
@pytest.mark.parametrize("nrow", [8_400_000, 10_000_000, 10_485_760])
def test_row_sums_queue_form_and_its_staged_stand_in(torch_cuda, pattern, nrow):
    """512-640 row blocks: the partition pass's queue form (whole aligned groups of 16 entries through one LDS queue per
    block; regions padded with entries of no row) on spread-out rows, and the staged form that stands by on the same
    padded layout when a cell of the count table says the rows are clustered.  Enough entries for several tiles per
    supertile and several supertiles; bit-stable, means through the parts' combine step."""
    torch = torch_cuda
    ncol, nnz = 500, 3_000_000
    rng = np.random.default_rng(len(pattern) + nrow % 97)
    x = rng.standard_normal(nnz)
    if pattern == "uniform":
        i = rng.integers(0, nrow, nnz)
    elif pattern == "one_row":
        i = np.full(nnz, nrow - 3)
    elif pattern == "sorted":
        i = np.sort(rng.integers(0, nrow, nnz))
    elif pattern == "edge_blocks":
        i = np.where(rng.random(nnz) < 0.5, rng.integers(0, 100, nnz), rng.integers(nrow - 100, nrow, nnz))
    elif pattern == "one_block_dense":
        i = rng.integers(300 * 16384, 301 * 16384, nnz)
    elif pattern == "all_invalid":
        i = np.full(nnz, -1)
    elif pattern == "first_tiles_invalid":
        i = rng.integers(0, nrow, nnz)
        i[:50_000] = nrow + 7
    else:
        i = np.where(rng.random(nnz) < 0.9, rng.integers(-nrow, 0, nnz), rng.integers(0, nrow, nnz))
    i = i.astype(np.int32)
    xt, it = torch.from_numpy(x).cuda(), torch.from_numpy(i).cuda()
    got = capi.row_sums_device(xt, it, nrow).cpu().numpy()
    again = capi.row_sums_device(xt, it, nrow).cpu().numpy()
    assert got.tobytes() == again.tobytes()
    keep = (i >= 0) & (i < nrow)
    ref = np.bincount(i[keep], weights=x[keep], minlength=nrow)
    scale = np.bincount(i[keep], weights=np.abs(x[keep]), minlength=nrow)
    assert np.all(np.abs(got - ref) <= RTOL * scale), float(np.max(np.abs(got - ref)))
    assert not np.any(np.signbit(got[scale == 0]))
    means = capi.row_sums_device(xt, it, nrow, ncol_for_means=ncol).cpu().numpy()
    assert means.tobytes() == (got / ncol).tobytes()


@pytest.mark.parametrize("pattern", ["one_row", "first_tiles_invalid", "edge_blocks", "one_block_dense", "all_invalid"])
def test_row_sums_skewed_tiles(torch_cuda, pattern):
    """Shapes of a partition tile the uniform generator never makes: every entry of a tile in ONE block
    (the sorted tile then takes three passes through the LDS stage and its runs are whole rounds), tiles
    without a single valid entry (which still have to fetch the next tile), entries only in the first and
    last block.  nrow gives 62 blocks, so the accumulate pass splits blocks among workgroups; means go
    through the parts' combine step as well."""
    torch = torch_cuda
    nrow, ncol, nnz = 1_000_000, 1_000, 300_000
    rng = np.random.default_rng(len(pattern))
    x = rng.standard_normal(nnz)
    if pattern == "one_row":
        i = np.full(nnz, 5, dtype=np.int32)
    elif pattern == "first_tiles_invalid":
        i = rng.integers(0, nrow, nnz).astype(np.int32)
        i[:50_000] = nrow + 7          # more than two tiles of 22528
    elif pattern == "edge_blocks":
        i = np.where(rng.random(nnz) < 0.5, rng.integers(0, 100, nnz), rng.integers(nrow - 100, nrow, nnz)).astype(np.int32)
    elif pattern == "one_block_dense":
        i = rng.integers(3 * 16384, 4 * 16384, nnz).astype(np.int32)
    else:
        i = np.full(nnz, -1, dtype=np.int32)
    xt, it = torch.from_numpy(x).cuda(), torch.from_numpy(i).cuda()
    got = capi.row_sums_device(xt, it, nrow).cpu().numpy()
    again = capi.row_sums_device(xt, it, nrow).cpu().numpy()
    assert got.tobytes() == again.tobytes()
    keep = (i >= 0) & (i < nrow)
    ref = np.bincount(i[keep], weights=x[keep], minlength=nrow)
    scale = np.bincount(i[keep], weights=np.abs(x[keep]), minlength=nrow)
    assert np.all(np.abs(got - ref) <= RTOL * scale), float(np.max(np.abs(got - ref)))
    assert not np.any(np.signbit(got[scale == 0]))
    means = capi.row_sums_device(xt, it, nrow, ncol_for_means=ncol).cpu().numpy()
    assert means.tobytes() == (got / ncol).tobytes()


@pytest.mark.parametrize("seed", range(25))
def test_fuzz_row_entries_and_masks(torch_cuda, seed):
    """Random shapes/densities: rowSums, rowMeans and the row-restricted column sums against the
    oracle, including matrices with empty rows/columns and a single row or column."""
    torch = torch_cuda
    rng = np.random.default_rng(500 + seed)
    nrow = int(rng.choice([1, 2, 7, 33, 500, 4096, 70_001]))
    ncol = int(rng.choice([1, 3, 64, 900, 12_345]))
    density = float(rng.choice([0.0005, 0.01, 0.2, 0.9]))
    nnz = min(int(nrow * ncol * density) + int(rng.integers(0, 3)), nrow * ncol)
    m = synth.rsparsematrix(nrow, ncol, nnz=nnz, seed=seed, kind=int(rng.integers(0, 2)))
    x, i, p = m["x"], m["i"], m["p"]
    xt, it, pt = torch.from_numpy(x).cuda(), torch.from_numpy(i).cuda(), torch.from_numpy(p).cuda()
    if nnz == 0:
        xt = torch.zeros(2, dtype=torch.float64, device="cuda")[:0]
        it = torch.zeros(2, dtype=torch.int32, device="cuda")[:0]
    check(capi.row_sums_device(xt, it, nrow).cpu().numpy(), x, i, p, nrow)
    s = np.flatnonzero(rng.random(nrow) < 0.4)
    bits = capi.row_set_bitmap(s, nrow)
    for comp in (False, True):
        got = capi.column_sums_in_rows_device(xt, it, pt, nrow, torch.from_numpy(bits).cuda(), comp).cpu().numpy()
        ref = oracle.column_sums_in_rows(x, i, p, bits, comp)
        keep = np.isin(i, s) != comp
        scale = oracle.column_abs_sums(np.where(keep, x, 0.0), p)
        assert np.all(np.abs(got - ref) <= RTOL * scale)


# ------------------------------------------------ the segments form behind a handle (no regrouped copy)
def _sorted_columns_matrix(nrow, ncol, mean, seed, invalid=False, duplicates=False):
    """CSC matrix whose columns' rows ascend: Poisson column lengths (some empty), uniformly drawn rows."""
    rng = np.random.default_rng(seed)
    counts = rng.poisson(mean, size=ncol).astype(np.int64)
    counts[rng.random(ncol) < 0.05] = 0
    col = np.repeat(np.arange(ncol, dtype=np.int64), counts)
    row = rng.integers(0, nrow, size=col.size, dtype=np.int64)
    if duplicates:
        row[1::7] = row[0::7][:row[1::7].size]                   # the same row twice in a column now and then
    if invalid:                                                  # rows outside [0, nrow): left out by every form
        row[rng.random(row.size) < 0.03] = nrow + 5
        row[rng.random(row.size) < 0.03] = -3
    order = np.lexsort((row, col))
    col, row = col[order], row[order]
    p = np.zeros(ncol + 1, dtype=np.int64)
    np.add.at(p, col + 1, 1)
    p = np.cumsum(p).astype(np.int32)
    x = synth.gen_values(int(p[-1]), seed=seed, kind=0)
    return x, row.astype(np.int32), p


def _handle_rows(x, i, p, nrow, ncol):
    h = capi.DeviceCSC(x, p, (nrow, ncol), i=i)
    assert h.row_form() == "none"
    hs, hs2, hm = h.row_sums(), h.row_sums(), h.row_means()
    form = h.row_form()
    cs = h.column_sums()
    h.close()
    assert hs.tobytes() == hs2.tobytes() and hm.tobytes() == (hs / ncol).tobytes()
    assert np.allclose(cs, oracle.column_sums(x, p), rtol=0, atol=1e-9)
    return hs, form


def test_handle_row_sums_take_the_segments_form_where_columns_are_long(torch_cuda):
    """40 000 rows (3 blocks of 16384), 600 columns of ~2000 entries: a column has ~670 entries per row block, the
    rows ascend -> no regrouping, a table of the columns' pieces per block (rsp_csc_row_form says "segments").
    Against the oracle's scatter loop; against the form the handle takes otherwise; matrices whose rows do not ascend,
    whose columns are short or which have a single row block keep the other forms."""
    nrow, ncol = 40_000, 600
    x, i, p = _sorted_columns_matrix(nrow, ncol, 2000, seed=3)
    hs, form = _handle_rows(x, i, p, nrow, ncol)
    assert form == "segments"
    check(hs, x, i, p, nrow)
    capi.set_row_segments(0)
    try:
        other, form0 = _handle_rows(x, i, p, nrow, ncol)
    finally:
        capi.set_row_segments(1)
    assert form0 == "direct"
    assert np.all(np.abs(hs - other) <= 2 * RTOL * row_l1(x, i, nrow))
    # rows that do not ascend inside a column: found by the check on the device, the handle regroups as before
    j0 = int(p[5]) + 10
    i2 = i.copy()
    i2[j0], i2[j0 + 1] = i[j0 + 1] + 1, i[j0]                      # (a descent in the middle of column 5)
    assert i2[j0] > i2[j0 + 1]
    hs2, form2 = _handle_rows(x, i2, p, nrow, ncol)
    assert form2 == "direct"
    check(hs2, x, i2, p, nrow)
    # short columns / one row block: not chosen
    xs, is_, ps = _sorted_columns_matrix(nrow, 5000, 40, seed=4)
    assert _handle_rows(xs, is_, ps, nrow, 5000)[1] == "direct"
    xb, ib, pb = _sorted_columns_matrix(16_384, 100, 3000, seed=5)
    assert _handle_rows(xb, ib, pb, 16_384, 100)[1] == "direct"


@pytest.mark.parametrize("nrow,ncol,mean,extra", [
    (16_385, 1, 700, ""), (16_385, 40, 0, ""), (40_000, 7, 300, ""), (40_000, 31, 64, "duplicates"),
    (70_001, 300, 500, "invalid"), (200_000, 90, 129, ""), (1_000_003, 64, 2000, ""), (1_000_003, 2000, 3, ""),
    (32_768, 15, 1000, ""), (32_769, 16, 1000, "invalid"),
])
def test_handle_row_sums_segments_form_forced_onto_small_and_odd_shapes(torch_cuda, nrow, ncol, mean, extra):
    """The segments form on shapes it would not choose (rsp_debug_set("row_segments", 2)): one column, fewer columns than
    staging wavefronts, empty matrices' worth of columns, a last row block of one row, 62 blocks with three entries
    per column, repeated rows, row indices outside [0, nrow) (left out, as in every form)."""
    x, i, p = _sorted_columns_matrix(nrow, ncol, mean, seed=nrow % 1000 + ncol, invalid=extra == "invalid",
                                     duplicates=extra == "duplicates")
    if len(x) == 0:
        x, i = np.array([2.5, -1.0]), np.array([3, nrow - 1], dtype=np.int32)
        p = np.zeros(ncol + 1, dtype=np.int32)
        p[1:] = 2
    capi.set_row_segments(2)
    try:
        hs, form = _handle_rows(x, i, p, nrow, ncol)
    finally:
        capi.set_row_segments(1)
    assert form == "segments"
    keep = (i >= 0) & (i < nrow)
    ref = oracle.row_sums(x[keep], i[keep], np.array([0, int(keep.sum())], dtype=np.int32), nrow)
    scale = np.bincount(i[keep], weights=np.abs(x[keep]), minlength=nrow)
    assert np.all(np.abs(hs - ref) <= RTOL * scale), float(np.max(np.abs(hs - ref) / np.maximum(scale, 1e-300)))
    assert np.all(hs[scale == 0] == 0.0) and not np.any(np.signbit(hs[scale == 0]))


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 4095, 4096, 4097, 8191, 1_000_003, 4096 * 1024, 4096 * 1025 + 7, (1 << 24) + 5])
def test_hand_written_exclusive_scan_of_the_count_tables(torch_cuda, n):
    """csrc/scan.hip (round 4: the last library call of the row-wise paths is gone): exclusive prefix sums of 32-bit counts,
    out of place and in place, from aligned and unaligned starts, against numpy -- exact (integer adds)."""
    torch = torch_cuda
    rng = np.random.default_rng(n)
    host = rng.integers(0, 120, size=n + 3, dtype=np.int32)
    if n > 100:
        host[rng.integers(0, n, size=5)] = 1_000_000          # a few large cells
    dev = torch.from_numpy(host).cuda()
    for off in (0, 3):                                       # (off = 3: the tile loads are not 16-byte aligned)
        src = dev[off:off + n]
        want = np.concatenate([[0], np.cumsum(host[off:off + n].astype(np.int64))[:-1]]).astype(np.int64)
        got = capi.exclusive_scan_device(src.contiguous() if off == 0 else src)      # a view at an odd offset stays a view
        assert np.array_equal(got.cpu().numpy().astype(np.int64), want), (n, off)
        buf = src.clone()
        capi.exclusive_scan_device(buf, buf)                 # in place
        assert np.array_equal(buf.cpu().numpy().astype(np.int64), want), (n, off, "in place")


def test_exclusive_scan_above_the_self_prefix_limit(torch_cuda):
    """scan.hip, round 5: up to 2048 tiles of 4096 counts every block of the last pass adds up the totals before it for
    itself; beyond that the one-block pass over the totals is back (the sizes above 8.4 million of the test above take it
    too).  One size well on the far side of the limit, in place and out of place."""
    torch = torch_cuda
    n = 4096 * 8193 + 11
    host = np.random.default_rng(5).integers(0, 40, size=n, dtype=np.int32)
    want = np.concatenate([[0], np.cumsum(host.astype(np.int64))[:-1]])
    buf = torch.from_numpy(host).cuda()
    got = capi.exclusive_scan_device(buf)
    assert np.array_equal(got.cpu().numpy().astype(np.int64), want)
    capi.exclusive_scan_device(buf, buf)
    assert np.array_equal(buf.cpu().numpy().astype(np.int64), want)
