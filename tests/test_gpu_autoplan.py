"""-m gpu: the plan-free device entries plan for themselves (round 5; round 6: plan on the second sighting, nothing ever
waits for the device, retired images are freed behind events, HBM use is bounded, rsp_column_sums_device_settle).

rsp_column_sums_device / rsp_column_means_device (include/rcppsparse_hip.h) remember the offsets they are shown: the
first call with (device, d_p, ncol, nnz) runs the general kernels and notes the key, the second enqueues a device-side
inspection of d_p behind them; once the host has seen its result, calls with that key take the lean form (every column short: one launch, the
reference's bits) or the columns form (every column long).  The caller promises nothing about d_p between calls, so
the kernels of this path check every column's offsets against the p[] of the call they run in -- what these tests are
mostly about: offsets changed in place under an adopted plan, captured calls replayed over changed offsets, two
matrices over two workspaces and streams, offsets that are not a dgCMatrix's at all.  Never a wrong sum, only a slower
call.  The reference has one synchronous call and no such state (src/example.cpp:26-32): the oracle is its loop."""
import os
import threading

import numpy as np
import pytest

import oracle
from rcppsparse_amd import capi, synth

pytestmark = pytest.mark.gpu
RTOL = 1e-12


@pytest.fixture(scope="module")
def torch_auto():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked tests need a GPU: the HIP path has no CPU fallback")
    capi.load()
    assert capi.debug_get("auto_plan") == 1      # the library's default, which the whole suite runs under
    capi.release_cached()
    yield torch
    capi.release_cached()


def check(got, x, p, exact=False):
    ref = oracle.column_sums(x, p)
    if exact:
        assert got.tobytes() == ref.tobytes()
        return
    scale = oracle.column_abs_sums(x, p)
    err = np.abs(got - ref)
    assert np.all(err <= RTOL * scale), float(np.max(err / np.maximum(scale, 1e-300)))
    empty = np.diff(p) == 0
    assert np.all(got[empty] == 0.0) and not np.any(np.signbit(got[empty]))


def short_matrix(ncol, mean, seed):
    """every column <= 64 entries (the lean form's shape: BASELINE config 2 is 1e6 columns of ~10)"""
    rng = np.random.default_rng(seed)
    counts = np.minimum(rng.poisson(mean, size=ncol), 64).astype(np.int64)
    counts[rng.integers(0, ncol, size=ncol // 50)] = 0
    p = synth.offsets_from_counts(counts)
    return p, synth.gen_values(int(p[-1]), seed=seed, kind=0)


def settled(torch, xt, pt, out, ws, stream=None):
    """a first call (general kernels), rsp_column_sums_device_settle, a call in the settled form; returns the form"""
    capi.column_sums_device(xt, pt, out, ws, stream=stream)
    form = capi.column_sums_device_settle(pt, xt.numel(), stream=stream)
    capi.column_sums_device(xt, pt, out, ws, stream=stream)
    torch.cuda.synchronize()
    return form


def test_c2_shaped_calls_settle_on_the_lean_form_with_the_references_bits(torch_auto):
    torch = torch_auto
    p, x = short_matrix(300_000, 10, seed=1)
    nnz = int(p[-1])
    assert nnz >= 2**20
    xt, pt = torch.from_numpy(x).cuda(), torch.from_numpy(p).cuda()
    out = torch.empty(len(p) - 1, dtype=torch.float64, device="cuda")
    ws = capi.alloc_workspace(len(p) - 1, nnz)
    assert capi.column_sums_device_form(pt, nnz) == "unknown"                  # never seen
    warm_p, warm_x = short_matrix(300_000, 10, seed=99)                        # (code objects, torch's staging pools: warm)
    capi.column_sums_device(torch.from_numpy(warm_x).cuda(), torch.from_numpy(warm_p).cuda()).cpu()
    torch.cuda.synchronize()
    made0 = capi.debug_get("auto_plans_made")
    free0 = torch.cuda.mem_get_info()[0]
    capi.column_sums_device(xt, pt, out, ws)
    first = out.cpu().numpy()
    check(first, x, p)                                                         # the general kernels answered the first call ...
    assert capi.column_sums_device_form(pt, nnz, wait=True) == "unknown"       # ... which only NOTED the key: no plan,
    assert torch.cuda.mem_get_info()[0] == free0 and capi.debug_get("auto_plans_made") == made0   # no allocation
    capi.column_sums_device(xt, pt, out, ws)                                   # second sighting: general kernels + the inspection
    assert out.cpu().numpy().tobytes() == first.tobytes()
    assert capi.column_sums_device_form(pt, nnz, wait=True) == "lean"
    capi.column_sums_device(xt, pt, out, ws)
    check(out.cpu().numpy(), x, p, exact=True)                                 # lean: every column in the reference's order
    # ... bit-identical to the form a caller gets by asking for a plan
    capi.set_lean(1)
    plan = capi.ColumnSumsPlan(p, nnz=nnz)
    assert plan.lean
    assert plan.column_sums(xt, pt).cpu().numpy().tobytes() == out.cpu().numpy().tobytes()
    plan.close()
    # another x behind the same offsets is the same key (bench.py rotates copies of x): still lean, still exact
    x2 = synth.gen_values(nnz, seed=77, kind=0)
    capi.column_sums_device(torch.from_numpy(x2).cuda(), pt, out, ws)
    check(out.cpu().numpy(), x2, p, exact=True)
    # colMeans through the same entry and the same plan (RcppSparse.h:145-150)
    means = capi.column_sums_device(xt, pt, nrow_for_means=12345).cpu().numpy()
    assert means.tobytes() == (oracle.column_sums(x, p) / 12345).tobytes()


def test_offsets_changed_in_place_under_an_adopted_plan_never_give_wrong_sums(torch_auto):
    """The same device buffers, new contents -- what a caching allocator or an in-place update of the matrix does.  The
    call right after the change runs the stale lean image: every changed column is caught by the kernel's own comparison and
    summed straight from x.  The call after that inspects again; then the new matrix has its own plan."""
    torch = torch_auto
    ncol = 250_000
    p, x = short_matrix(ncol, 9, seed=2)
    nnz = int(p[-1])
    xt, pt = torch.from_numpy(x).cuda(), torch.from_numpy(p).cuda()
    out = torch.empty(ncol, dtype=torch.float64, device="cuda")
    ws = capi.alloc_workspace(ncol, nnz)
    assert settled(torch, xt, pt, out, ws) == "lean"
    rng = np.random.default_rng(3)
    variants = []
    # (a) a few boundaries moved by one or two entries; (b) another matrix altogether, same ncol and nnz;
    # (c) one giant column swallowing a third of x and a run of empty columns; (d) everything in the last column
    a = p.copy()
    for c in rng.integers(1, ncol - 1, size=200):
        if a[c - 1] < a[c] < a[c + 1]:
            a[c] += int(rng.integers(-1, 2))
    variants.append(a)
    counts = rng.multinomial(nnz, np.full(ncol, 1.0 / ncol)).astype(np.int64)
    variants.append(synth.offsets_from_counts(counts))
    g = np.zeros(ncol, dtype=np.int64)
    g[1000] = nnz // 3
    rest = nnz - nnz // 3
    g[50_000:50_000 + rest // 7] = 7
    g[ncol - 1] += nnz - int(g.sum())
    variants.append(synth.offsets_from_counts(g))
    last = np.zeros(ncol + 1, dtype=np.int32)
    last[-1] = nnz
    variants.append(last)
    for q in variants:
        q = np.ascontiguousarray(q, dtype=np.int32)
        assert q[0] == 0 and q[-1] == nnz and np.all(np.diff(q) >= 0)
        pt.copy_(torch.from_numpy(q))                                     # in place: same address, same sizes
        torch.cuda.synchronize()
        for _ in range(4):                                                # stale image -> re-inspection -> new plan (or none)
            out.fill_(-1.0)
            capi.column_sums_device(xt, pt, out, ws)
            check(out.cpu().numpy(), x, q)
        form = capi.column_sums_device_settle(pt, nnz)
        assert form in ("lean", "general", "columns")
        out.fill_(-1.0)
        capi.column_sums_device(xt, pt, out, ws)
        check(out.cpu().numpy(), x, q, exact=(form == "lean"))


def test_a_capture_records_the_general_kernels_and_owns_nothing_of_the_library(torch_auto):
    """A HIP graph outlives the call that was captured, and the images of the entry's own plans belong to the library (an
    eviction or rsp_release_cached frees them): a call on a capturing stream therefore records the general kernels whatever
    is known about the offsets.  The graph keeps working after rsp_release_cached and over new values AND new offsets."""
    torch = torch_auto
    ncol = 200_000
    p, x = short_matrix(ncol, 11, seed=4)
    nnz = int(p[-1])
    xt, pt = torch.from_numpy(x).cuda(), torch.from_numpy(p).cuda()
    out = torch.empty(ncol, dtype=torch.float64, device="cuda")
    ws = capi.alloc_workspace(ncol, nnz)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        assert settled(torch, xt, pt, out, ws) == "lean"
        planned_bits = out.cpu().numpy().tobytes()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            capi.column_sums_device(xt, pt, out, ws)
        capi.release_cached()                                              # everything the library kept is gone ...
        out.fill_(-1.0)
        g.replay()                                                         # ... and the graph does not care
        torch.cuda.synchronize()
        check(out.cpu().numpy(), x, p)
        capi.set_auto_plan(False)
        general_bits = capi.column_sums_device(xt, pt).cpu().numpy().tobytes()
        capi.set_auto_plan(True)
        assert out.cpu().numpy().tobytes() == general_bits                 # what was recorded is the general kernels
        assert planned_bits == oracle.column_sums(x, p).tobytes()          # (the eager call before it had the lean plan's bits)
        p2, _ = short_matrix(ncol, 11, seed=5)
        p2 = np.minimum(p2, nnz).astype(np.int32)
        p2[-1] = nnz
        x2 = synth.gen_values(nnz, seed=55, kind=0)
        xt.copy_(torch.from_numpy(x2))
        pt.copy_(torch.from_numpy(p2))
        out.fill_(-1.0)
        g.replay()
        torch.cuda.synchronize()
        check(out.cpu().numpy(), x2, p2)
    # to have the PLANNED form in a graph the caller makes the plan and owns it (rsp_column_sums_plan_*):
    plan = capi.ColumnSumsPlan(pt, nnz=nnz, stream=s).wait()
    with torch.cuda.stream(s):
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2, stream=s):
            plan.column_sums(xt, pt, out, ws)
        out.fill_(-1.0)
        g2.replay()
        torch.cuda.synchronize()
    check(out.cpu().numpy(), x2, p2, exact=plan.lean)
    plan.close()


def test_two_matrices_two_workspaces_two_streams_interleaved(torch_auto):
    torch = torch_auto
    mats = []
    for seed, mean in ((7, 8), (8, 14)):
        p, x = short_matrix(220_000, mean, seed=seed)
        mats.append((p, x, torch.from_numpy(x).cuda(), torch.from_numpy(p).cuda()))
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    wss = [capi.alloc_workspace(220_000, max(len(m[1]) for m in mats)) for _ in range(2)]
    outs = [[torch.empty(220_000, dtype=torch.float64, device="cuda") for _ in range(2)] for _ in range(2)]
    torch.cuda.synchronize()
    for rnd in range(12):                                                  # matrix m over workspace / stream (m + rnd) % 2
        for m, (p, x, xt, pt) in enumerate(mats):
            k = (m + rnd) % 2
            capi.column_sums_device(xt, pt, outs[m][k], wss[k], stream=streams[k])
        if rnd == 2:
            for m, (p, x, xt, pt) in enumerate(mats):
                assert capi.column_sums_device_form(pt, len(x), wait=True) == "lean"
        torch.cuda.synchronize()
        for m, (p, x, xt, pt) in enumerate(mats):
            check(outs[m][(m + rnd) % 2].cpu().numpy(), x, p, exact=rnd > 2)


def test_long_columns_settle_on_the_columns_form_and_notice_a_different_matrix(torch_auto):
    torch = torch_auto
    ncol = 1000
    rng = np.random.default_rng(9)
    counts = rng.integers(3000, 5200, size=ncol).astype(np.int64)          # the reference vignette's shape, smaller
    p = synth.offsets_from_counts(counts)
    x = synth.gen_values(int(p[-1]), seed=9, kind=0)
    nnz = int(p[-1])
    xt, pt = torch.from_numpy(x).cuda(), torch.from_numpy(p).cuda()
    out = torch.empty(ncol, dtype=torch.float64, device="cuda")
    ws = capi.alloc_workspace(ncol, nnz)
    assert settled(torch, xt, pt, out, ws) == "columns"
    check(out.cpu().numpy(), x, p)
    again = out.clone()
    capi.column_sums_device(xt, pt, out, ws)
    assert torch.equal(out, again)                                        # settled: bit-stable from here on
    # the same buffers now hold a matrix of 990 empty columns and 10 giant ones: right at once, re-inspected afterwards
    g = np.zeros(ncol, dtype=np.int64)
    g[::100] = nnz // 10
    g[-1] += nnz - int(g.sum())
    q = synth.offsets_from_counts(g)
    pt.copy_(torch.from_numpy(q))
    for _ in range(3):
        out.fill_(-1.0)
        capi.column_sums_device(xt, pt, out, ws)
        check(out.cpu().numpy(), x, q)
    assert capi.column_sums_device_settle(pt, nnz) in ("general", "columns")
    capi.column_sums_device(xt, pt, out, ws)
    check(out.cpu().numpy(), x, q)


def test_offsets_that_are_no_dgcmatrix_stay_in_bounds_under_an_adopted_plan(torch_auto):
    """The device entries trust the caller's p[] for the VALUES of the sums only: reads and writes stay in bounds for any
    content (include/rcppsparse_hip.h) -- also when a lean or columns plan made from valid offsets is in force."""
    torch = torch_auto
    rng = np.random.default_rng(10)
    # (offsets that are garbage name garbage ranges: a call may then be slow -- every column is summed over whatever its two
    # offsets say -- so the lean case stays small here)
    for make, want in ((lambda: short_matrix(120_000, 10, seed=11), "lean"),
                       (lambda: (synth.offsets_from_counts(np.full(600, 4000, dtype=np.int64)),
                                 synth.gen_values(600 * 4000, seed=12, kind=0)), "columns")):
        p, x = make()
        ncol, nnz = len(p) - 1, int(p[-1])
        xt, pt = torch.from_numpy(x).cuda(), torch.from_numpy(p).cuda()
        guard = torch.full((ncol + 64,), 7.0, dtype=torch.float64, device="cuda")
        out = guard[32:32 + ncol]
        ws = capi.alloc_workspace(ncol, nnz)
        assert settled(torch, xt, pt, out, ws) == want
        for bad in (rng.integers(-2**31, 2**31 - 1, size=ncol + 1), np.full(ncol + 1, nnz + 5), np.arange(ncol + 1)[::-1] * 3):
            pt.copy_(torch.from_numpy(np.ascontiguousarray(bad, dtype=np.int64).astype(np.int32)))
            capi.column_sums_device(xt, pt, out, ws)
            torch.cuda.synchronize()                                      # no fault ...
            assert bool(torch.all(guard[:32] == 7.0)) and bool(torch.all(guard[32 + ncol:] == 7.0))   # ... nothing outside the result
        pt.copy_(torch.from_numpy(p))
        for _ in range(3):
            capi.column_sums_device(xt, pt, out, ws)
        torch.cuda.synchronize()
        check(out.cpu().numpy(), x, p)


def test_infinities_and_na_go_through_the_changed_column_fall_back_like_a_plain_add(torch_auto):
    """The fall-back that sums a changed column straight from x is compensated (a column of any length may stand there); an
    infinity, a NaN or R's NA_real_ must still come out as the reference's plain += carries them."""
    torch = torch_auto
    ncol = 150_000
    p, x = short_matrix(ncol, 10, seed=40)
    nnz = int(p[-1])
    x = x.copy()
    na = np.frombuffer(np.array([0x7FF00000000007A2], dtype=np.uint64).tobytes(), dtype=np.float64)[0]
    rng = np.random.default_rng(41)
    q = synth.offsets_from_counts(rng.multinomial(nnz, np.full(ncol, 1.0 / ncol)).astype(np.int64))
    long_cols = np.flatnonzero(np.diff(q) >= 6)[:400:40]                  # ten columns of the NEW matrix with room for specials
    kinds = []
    for k, c in enumerate(long_cols):
        a = int(q[c])
        if k % 5 == 0:
            x[a + 2] = na; kinds.append("na")
        elif k % 5 == 1:
            x[a + 1] = np.inf; kinds.append("inf")
        elif k % 5 == 2:
            x[a] = -np.inf; x[a + 3] = 7.0; kinds.append("-inf")
        elif k % 5 == 3:
            x[a + 1] = np.inf; x[a + 4] = -np.inf; kinds.append("nan")
        else:
            x[a + 5] = na; x[a] = np.inf; kinds.append("na")
    xt, pt = torch.from_numpy(x).cuda(), torch.from_numpy(p).cuda()
    out = torch.empty(ncol, dtype=torch.float64, device="cuda")
    ws = capi.alloc_workspace(ncol, nnz)
    assert settled(torch, xt, pt, out, ws) == "lean"
    pt.copy_(torch.from_numpy(q))                                         # every column now differs from the image: all through the fall-back
    capi.column_sums_device(xt, pt, out, ws)
    got = out.cpu().numpy()
    ref = oracle.column_sums(x, q)
    finite = np.isfinite(ref)
    scale = oracle.column_abs_sums(x, q)
    assert np.all(np.abs(got[finite] - ref[finite]) <= RTOL * scale[finite])
    for c, kind in zip(long_cols, kinds):
        if kind == "na":
            assert int(got[c:c + 1].view(np.uint64)[0]) == 0x7FF80000000007A2 == int(ref[c:c + 1].view(np.uint64)[0])
        elif kind == "nan":
            assert np.isnan(got[c]) and np.isnan(ref[c])
        else:
            assert got[c] == ref[c] == (np.inf if kind == "inf" else -np.inf)


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_adopted_forms_against_the_oracle(torch_auto, seed):
    """Random shapes on both sides of the forms' conditions, each summed before the plan is known, after it, and after the
    offsets were replaced in place by another random matrix of the same sizes."""
    torch = torch_auto
    rng = np.random.default_rng(100 + seed)
    kind = seed % 4
    if kind == 0:        # short columns, some chunks crowded
        mean = rng.uniform(3, 30)
        ncol = max(int(rng.integers(120_000, 400_000)), int(1.4e6 / mean))
        counts = np.minimum(rng.poisson(mean, size=ncol), 64).astype(np.int64)
    elif kind == 1:      # short columns with long runs of empty ones
        ncol = int(rng.integers(200_000, 500_000))
        counts = np.where(rng.random(ncol) < 0.3, rng.integers(1, 65, size=ncol), 0).astype(np.int64)
    elif kind == 2:      # long similar columns
        ncol = int(rng.integers(300, 900))
        counts = rng.integers(4000, 9000, size=ncol).astype(np.int64)
    else:                # Zipf: neither form
        ncol = int(rng.integers(5_000, 50_000))
        counts = synth.zipf_counts(ncol, 2_000_000, seed=seed, nrow=400_000)
    p = synth.offsets_from_counts(counts)
    nnz = int(p[-1])
    if nnz < 2**20:
        pytest.skip("below the size the entry plans for")
    x = synth.gen_values(nnz, seed=seed, kind=seed % 2)
    xt, pt = torch.from_numpy(x).cuda(), torch.from_numpy(p).cuda()
    out = torch.empty(ncol, dtype=torch.float64, device="cuda")
    ws = capi.alloc_workspace(ncol, nnz)
    capi.column_sums_device(xt, pt, out, ws)
    check(out.cpu().numpy(), x, p)
    form = capi.column_sums_device_settle(pt, nnz)
    assert form in ("lean", "columns", "general")
    if kind == 3:
        assert form == "general"
    for _ in range(2):
        capi.column_sums_device(xt, pt, out, ws)
    check(out.cpu().numpy(), x, p, exact=(form == "lean"))
    q = synth.offsets_from_counts(rng.multinomial(nnz, rng.dirichlet(np.full(ncol, 0.7))).astype(np.int64))
    pt.copy_(torch.from_numpy(q))
    for _ in range(3):
        out.fill_(-1.0)
        capi.column_sums_device(xt, pt, out, ws)
        check(out.cpu().numpy(), x, q)


def test_the_entrys_own_lean_plan_at_the_int32_limit(torch_auto):
    """nnz = 2^31 - 1, the most the reference's 32-bit p[] can address (RcppSparse.h:30), in 2.1e8 columns of ten entries: the
    device-side inspection of 860 MB of offsets, a lean image of 4.2e6 chunks, and the validating kernel's comparisons next
    to 2^31 (chunk start + offset passes INT32_MAX in the last chunks).  Entries are small integers, so the expected sums
    are exact whatever the order and come from integer arithmetic alone."""
    torch = torch_auto
    if torch.cuda.get_device_properties(0).total_memory < 48 * 2**30:
        pytest.skip("needs >= 48 GB of HBM")
    try:
        free_host = os.sysconf("SC_AVPHYS_PAGES") * os.sysconf("SC_PAGE_SIZE")
    except (ValueError, OSError):
        free_host = 0
    if free_host < 12 * 2**30:
        pytest.skip("needs >= 12 GB of free host memory")
    torch.cuda.empty_cache()
    nnz, per = 2**31 - 1, 10
    ncol = -(-nnz // per)
    f = lambda idx: ((idx * 2654435761) >> 13) % 7 - 3   # noqa: E731  (the same few integer operations on host and device)
    want = np.empty(ncol, dtype=np.float64)
    step = 50_000_000
    for s0 in range(0, nnz, step):
        n = min(step, nnz - s0)
        v = f(np.arange(s0, s0 + n, dtype=np.int64))
        full = n // per
        want[s0 // per:s0 // per + full] = v[:full * per].reshape(full, per).sum(axis=1)
        if full * per < n:
            want[s0 // per + full] = v[full * per:].sum()
    xt = torch.empty(nnz, dtype=torch.float64, device="cuda")
    for s0 in range(0, nnz, 100_000_000):
        n = min(100_000_000, nnz - s0)
        xt[s0:s0 + n] = f(torch.arange(s0, s0 + n, dtype=torch.int64, device="cuda")).double()
    pt = torch.clamp(torch.arange(0, ncol + 1, dtype=torch.int64, device="cuda") * per, max=nnz).to(torch.int32)
    assert int(pt[-1]) == nnz and int(pt[-2]) == (ncol - 1) * per
    out = torch.empty(ncol, dtype=torch.float64, device="cuda")
    ws = capi.alloc_workspace(ncol, nnz)
    capi.column_sums_device(xt, pt, out, ws)                              # the general kernels
    assert out.cpu().numpy().tobytes() == (want + 0.0).tobytes()
    assert capi.column_sums_device_settle(pt, nnz) == "lean"              # the inspection of 860 MB of offsets, waited for
    out.fill_(-1.0)
    capi.column_sums_device(xt, pt, out, ws)
    assert out.cpu().numpy().tobytes() == (want + 0.0).tobytes()
    # the last columns moved by one entry, in place: caught and summed straight from x, right next to 2^31
    q = pt.clone()
    q[-3] -= 1
    q[-2] += 2
    pt.copy_(q)
    want2 = want.copy()
    tail = f(np.arange(nnz - 40, nnz, dtype=np.int64))
    qh = q[-5:].cpu().numpy().astype(np.int64)
    for k in range(4):
        want2[ncol - 4 + k] = tail[qh[k] - (nnz - 40):qh[k + 1] - (nnz - 40)].sum()
    out.fill_(-1.0)
    capi.column_sums_device(xt, pt, out, ws)
    assert out.cpu().numpy().tobytes() == (want2 + 0.0).tobytes()
    capi.release_cached()


def test_auto_plan_off_keeps_every_call_on_the_general_kernels(torch_auto):
    torch = torch_auto
    p, x = short_matrix(200_000, 10, seed=20)
    nnz = int(p[-1])
    xt, pt = torch.from_numpy(x).cuda(), torch.from_numpy(p).cuda()
    capi.set_auto_plan(False)
    try:
        assert capi.debug_get("auto_plan") == 0
        first = capi.column_sums_device(xt, pt).cpu().numpy().tobytes()
        for _ in range(5):
            assert capi.column_sums_device(xt, pt).cpu().numpy().tobytes() == first     # bit-stable from the first call
        assert capi.column_sums_device_form(pt, nnz) == "unknown"
    finally:
        capi.set_auto_plan(True)
    assert capi.debug_get("auto_plan") == 1


def test_more_keys_than_the_library_remembers(torch_auto):
    """16 keys are remembered.  Keys that never got a plan make room first (they hold nothing); a PLANNED key only after 64
    calls without a use, and then its plan is retired behind an event -- nothing waits for the device."""
    torch = torch_auto
    capi.release_cached()
    keep = []
    for k in range(20):
        p, x = short_matrix(110_000, 10, seed=30 + k)
        xt, pt = torch.from_numpy(x).cuda(), torch.from_numpy(p).cuda()
        keep.append((p, x, xt, pt))
    torch.cuda.synchronize()
    for p, x, xt, pt in keep:                                             # two sightings each: the first 16 get their plan,
        for _ in range(2):                                                # the other four find the table full of keys in use
            check(capi.column_sums_device(xt, pt).cpu().numpy(), x, p)
    forms = [capi.column_sums_device_form(pt, len(x), wait=True) for p, x, xt, pt in keep]
    assert forms[:16] == ["lean"] * 16 and forms[16:] == ["unknown"] * 4
    for p, x, xt, pt in keep:
        check(capi.column_sums_device(xt, pt).cpu().numpy(), x, p, exact=None)
    # key 0 used 70 more times: the other fifteen have gone idle, and key 16 now takes the place of the longest-idle one
    p0, x0, xt0, pt0 = keep[0]
    for _ in range(70):
        capi.column_sums_device(xt0, pt0)
    p16, x16, xt16, pt16 = keep[16]
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(2):
        check(capi.column_sums_device(xt16, pt16).cpu().numpy(), x16, p16)
    assert capi.column_sums_device_form(pt16, len(x16), wait=True) == "lean"
    assert capi.column_sums_device_form(keep[1][3], len(keep[1][1])) == "unknown"     # key 1 made room ...
    assert capi.column_sums_device_form(pt0, len(x0)) == "lean"                          # ... key 0, in use, did not
    torch.cuda.synchronize()
    check(capi.column_sums_device(xt16, pt16).cpu().numpy(), x16, p16, exact=True)      # (this call frees key 1's retired image)
    assert abs(torch.cuda.mem_get_info()[0] - free0) < 2**22                             # one image came, one went
    capi.release_cached()
    assert capi.column_sums_device_form(keep[0][3], len(keep[0][1])) == "unknown"


def test_a_stream_of_fresh_keys_allocates_nothing_and_never_waits(torch_auto):
    """fresh offsets in every call (a new p buffer per batch) must not cost an allocation, an inspection or
    a device synchronisation.  200 unique keys, one call each: free HBM does not move, and a deliberately slow kernel
    queued on another stream is still running when the calls have returned (nothing waited for the device)."""
    torch = torch_auto
    capi.release_cached()
    p, x = short_matrix(120_000, 10, seed=60)
    nnz, ncol = int(p[-1]), len(p) - 1
    xt = torch.from_numpy(x).cuda()
    pts = [torch.from_numpy(p).cuda() for _ in range(200)]               # 200 different addresses: 200 keys
    out = torch.empty(ncol, dtype=torch.float64, device="cuda")
    ws = capi.alloc_workspace(ncol, nnz)
    big = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        big.add_(1.0)                                                      # (its code object loaded before anything is measured)
    torch.cuda.synchronize()
    free0, made0 = torch.cuda.mem_get_info()[0], capi.debug_get("auto_plans_made")
    with torch.cuda.stream(side):
        for _ in range(40):
            big.add_(1.0)                                                  # ~40 x 2 GB of traffic: several milliseconds
        done = torch.cuda.Event()
        done.record(side)
    for pt in pts:
        capi.column_sums_device(xt, pt, out, ws)
    still_running = not done.query()
    torch.cuda.synchronize()
    assert capi.debug_get("auto_plans_made") == made0 and abs(torch.cuda.mem_get_info()[0] - free0) <= 2**22
    assert still_running, "the calls outlasted several milliseconds of queued work: something waited for the device"
    check(out.cpu().numpy(), x, p)


def test_rewriting_the_offsets_in_place_keeps_hbm_bounded_and_the_key_planned(torch_auto):
    """VERDICT round 5, next 4 / ADVICE: a caller that rewrites p[] in place every ~70 calls used to leak one image per
    rewrite.  600 rewrites x 70 calls on one stream: every rewrite retires a plan; its image is freed once the event
    recorded behind the retiring call has completed; free HBM stays within two images of the start, and the key is still
    planned at the end (32 clean calls forgive the strike)."""
    torch = torch_auto
    capi.release_cached()
    ncol = 150_000
    pa, x = short_matrix(ncol, 10, seed=70)
    nnz = int(pa[-1])
    rng = np.random.default_rng(71)
    pb = synth.offsets_from_counts(np.minimum(rng.multinomial(nnz, np.full(ncol, 1.0 / ncol)), 64).astype(np.int64))
    if int(pb[-1]) != nnz:                                                 # (the clamp to 64 moved entries: spread the rest)
        pb = pa.copy()
        pb[1:-1] = np.minimum(pb[1:-1] + 1, pb[2:])
    qa, qb = torch.from_numpy(pa).cuda(), torch.from_numpy(np.ascontiguousarray(pb, dtype=np.int32)).cuda()
    xt, pt = torch.from_numpy(x).cuda(), torch.from_numpy(pa).cuda()
    out = torch.empty(ncol, dtype=torch.float64, device="cuda")
    ws = capi.alloc_workspace(ncol, nnz)
    capi.column_sums_device(xt, pt, out, ws)
    assert capi.column_sums_device_settle(pt, nnz) == "lean"
    image = torch.cuda.mem_get_info()[0]
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    made0 = capi.debug_get("auto_plans_made")
    for r in range(600):
        pt.copy_(qb if r % 2 == 0 else qa)                                 # in place, on the calls' stream ...
        torch.cuda.synchronize()                                           # ... and waited for, as a synchronous caller (R) does
        for _ in range(70):
            capi.column_sums_device(xt, pt, out, ws)
        if r % 100 == 99:
            torch.cuda.synchronize()
            check(out.cpu().numpy(), x, pb if r % 2 == 0 else pa)
            assert abs(torch.cuda.mem_get_info()[0] - free0) <= 3 * 2**21, (r, free0 - torch.cuda.mem_get_info()[0])
    torch.cuda.synchronize()
    assert capi.column_sums_device_settle(pt, nnz) == "lean"              # not given up on
    capi.column_sums_device(xt, pt, out, ws)
    check(out.cpu().numpy(), x, pa, exact=True)
    assert capi.debug_get("auto_plans_made") - made0 >= 300               # (the key really was re-planned all along)
    assert capi.debug_get("auto_plans_retired") <= 2
    assert capi.debug_get("auto_plans_recycled") >= 290                   # ... out of recycled allocations: nothing was freed
    # and none of it waits for the device: freeing device or page-locked memory drains EVERY stream on this runtime (21 ms
    # behind 21 ms of queued work), so retired plans' allocations are recycled, never freed, inside a call.  Two more
    # rewrite cycles with a slow kernel queued on another stream: it is still running when the cycles' calls have returned.
    big = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        big.add_(1.0)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(60):
            big.add_(1.0)
        done = torch.cuda.Event()
        done.record(side)
    for r in range(2):
        pt.copy_(qb if r % 2 == 0 else qa)
        torch.cuda.current_stream().synchronize()                          # (the caller waits for ITS stream only)
        for _ in range(70):
            capi.column_sums_device(xt, pt, out, ws)
        torch.cuda.current_stream().synchronize()
    assert not done.query(), "a call waited for the device (something was freed inside it)"
    torch.cuda.synchronize()
    check(out.cpu().numpy(), x, pa, exact=None)
    del image, big


def test_a_deep_queue_behind_one_in_place_update_costs_one_strike_not_the_key(torch_auto):
    """with ONE stale word shared by a key's plans, launches of the old image that were already queued raised
    it again after the host had reset it, and a single legitimate update could use up every strike.  Every plan now has its
    own word: 200 calls queued without a synchronisation right behind an update retire ONE plan, and the key is lean again."""
    torch = torch_auto
    capi.release_cached()
    ncol = 200_000
    pa, x = short_matrix(ncol, 9, seed=80)
    nnz = int(pa[-1])
    pb = pa.copy()
    inner = np.flatnonzero((pa[1:-1] > pa[:-2]) & (pa[1:-1] < pa[2:]))[::7] + 1
    pb[inner] -= 1                                                         # thousands of boundaries moved by one entry
    xt, pt = torch.from_numpy(x).cuda(), torch.from_numpy(pa).cuda()
    qb = torch.from_numpy(pb).cuda()
    out = torch.empty(ncol, dtype=torch.float64, device="cuda")
    ws = capi.alloc_workspace(ncol, nnz)
    capi.column_sums_device(xt, pt, out, ws)
    assert capi.column_sums_device_settle(pt, nnz) == "lean"
    for rnd in range(6):                                                   # six updates: more than kAutoMaxStrikes, were they double-counted
        pt.copy_(qb if rnd % 2 == 0 else torch.from_numpy(pa).cuda())
        for _ in range(200):
            capi.column_sums_device(xt, pt, out, ws)                       # (no synchronisation: a deep queue)
        torch.cuda.synchronize()
        check(out.cpu().numpy(), x, pb if rnd % 2 == 0 else pa)
    assert capi.column_sums_device_form(pt, nnz, wait=True) == "lean"
    capi.column_sums_device(xt, pt, out, ws)
    check(out.cpu().numpy(), x, pa, exact=True)


def test_bit_stable_run_to_run_under_the_defaults(torch_auto):
    """include/rcppsparse_hip.h: from the return of rsp_column_sums_device_settle on, every call with the key returns the
    same bits -- two streams, sums and means, fifty calls."""
    torch = torch_auto
    capi.release_cached()
    p, x = short_matrix(260_000, 12, seed=90)
    nnz, ncol = int(p[-1]), len(p) - 1
    xt, pt = torch.from_numpy(x).cuda(), torch.from_numpy(p).cuda()
    assert capi.column_sums_device_settle(pt, nnz) == "lean"              # (no call before it: settle makes the plan itself)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    wss = [capi.alloc_workspace(ncol, nnz) for _ in streams]
    outs = [torch.empty(ncol, dtype=torch.float64, device="cuda") for _ in streams]
    want = oracle.column_sums(x, p).tobytes()
    for k in range(50):
        j = k % 2
        capi.column_sums_device(xt, pt, outs[j], wss[j], stream=streams[j])
        if k % 10 >= 8:
            streams[j].synchronize()
            assert outs[j].cpu().numpy().tobytes() == want
    torch.cuda.synchronize()
    assert capi.column_sums_device(xt, pt, nrow_for_means=777).cpu().numpy().tobytes() == (oracle.column_sums(x, p) / 777).tobytes()


def test_planning_on_one_thread_leaves_another_threads_global_capture_alone(torch_auto):
    """an allocation, a page-lock or an event query on ANY thread invalidates a stream capture that another
    thread runs in GLOBAL mode -- and a key's second sighting allocates its plan.  The entries make those few calls with their
    own thread's capture mode relaxed (hipThreadExchangeStreamCaptureMode): a capture in global mode on another thread
    survives a first and a second sighting, the settling of another key and a stale round on this thread, and replays."""
    torch = torch_auto
    capi.release_cached()
    mats = []
    for seed in (95, 96):
        p, x = short_matrix(150_000, 10, seed=seed)
        mats.append((p, x, torch.from_numpy(x).cuda(), torch.from_numpy(p).cuda(),
                     torch.empty(len(p) - 1, dtype=torch.float64, device="cuda"), capi.alloc_workspace(len(p) - 1, len(x))))
    a = torch.zeros(4096, device="cuda")
    cap_stream, my_stream = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    started, go_on, errors = threading.Event(), threading.Event(), []

    def capturer():
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=cap_stream, capture_error_mode="global"):
                a.add_(1.0)
                started.set()
                go_on.wait(30)
                a.add_(1.0)
            g.replay()
            torch.cuda.synchronize()
        except Exception as e:   # noqa: BLE001
            errors.append(repr(e))
            started.set()

    t = threading.Thread(target=capturer)
    t.start()
    assert started.wait(30) and not errors, errors
    made0 = capi.debug_get("auto_plans_made")
    with torch.cuda.stream(my_stream):                       # (no work on the legacy stream, no torch allocation: those would
        p, x, xt, pt, out, ws = mats[0]                      #  disturb the capture by themselves)
        for _ in range(3):                                   # first sighting, second (allocates + inspects), third
            capi.column_sums_device(xt, pt, out, ws, stream=my_stream)
        p2, x2, xt2, pt2, out2, ws2 = mats[1]
        assert capi.column_sums_device_settle(pt2, len(x2), stream=my_stream) == "lean"
        pt.copy_(pt2[:pt.numel()] if pt2.numel() >= pt.numel() else pt)      # offsets changed in place: a stale round, a retirement
        for _ in range(4):
            capi.column_sums_device(xt, pt, out, ws, stream=my_stream)
        # (no synchronisation from THIS thread while the other one captures: a stream wait is itself one of the forbidden calls)
    assert capi.debug_get("auto_plans_made") - made0 >= 2
    go_on.set()
    t.join(60)
    assert not errors, errors
    torch.cuda.synchronize()
    assert float(a[0]) == 2.0                                # the two captured adds, replayed once
