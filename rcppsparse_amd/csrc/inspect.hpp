// inspect.hpp -- the inspectors of the inspector-executor forms of columnSums: pure host
// C++ over a host copy of p[] (reference inst/include/RcppSparse.h:220-221: column c is [p[c], p[c+1])), no HIP
// types, so that the same code that capi.hip ships is also built on the CPU with the address and undefined-behaviour
// sanitizers and checked against naive restatements (tests/c/inspect_selftest.cpp, tests/test_plan_inspector.py).
#ifndef RSP_INSPECT_HPP
#define RSP_INSPECT_HPP

#include <algorithm>
#include <atomic>
#include <climits>
#include <cstdint>
#include <thread>
#include <vector>

// The few index functions below are shared with the device-side inspector (inspect_device.hip), which runs the same
// arithmetic with a thread per column: they are plain integer code, marked for both sides when hipcc compiles them.
#ifdef __HIPCC__
#define RSP_INSPECT_HD __host__ __device__
#else
#define RSP_INSPECT_HD
#endif

namespace rsp {
namespace inspect {

struct Rec {   // two 32-bit numbers per chunk (the device sees them as int2)
    int32_t a, b;
};

// The chunk grid of a call: `nbody` chunks of `body` elements, then chunks of `tail` elements (colsums_kernels.h ChunkMap).
struct Grid {
    int32_t body, nbody, tail, nchunks;
    RSP_INSPECT_HD int64_t start(int32_t w) const {
        return w < nbody ? (int64_t)w * body : (int64_t)nbody * body + (int64_t)(w - nbody) * tail;
    }
    RSP_INSPECT_HD int32_t chunk_of(int64_t e) const {   // the chunk whose grid range holds element e (may be >= nchunks past the end)
        const int64_t edge = (int64_t)nbody * body;
        return e < edge ? (int32_t)(e / body) : nbody + (int32_t)((e - edge) / tail);
    }
};

// ---- the inspection as ONE pass over p[] (what the device runs, a thread per column index) ------------------
// Column index c (0..ncol, p[ncol] closing) is the first column start at or after a grid position g exactly when
// p[c-1] < g <= p[c] (p[-1] = -1).  So the index that sees p[c-1] < p[c] answers for the chunks lo..hi below by two
// divisions instead of every chunk searching for its column.
struct Span {
    int32_t lo, hi;   // chunks lo..hi; empty when lo > hi
};
RSP_INSPECT_HD inline Span span_of(const Grid& g, int32_t prev, int32_t v) {
    Span s;
    s.lo = prev < 0 ? 0 : g.chunk_of(prev) + 1;
    s.hi = g.chunk_of(v);                              // (start(hi) <= v)
    if (s.hi > g.nchunks - 1) s.hi = g.nchunks - 1;    // (v == nnz may sit on the grid's closing edge)
    return s;
}
// last index e in [c, ncol] with p[e] == v, given p[c] == v (a run of empty columns behind column c): gallop, then bisect
RSP_INSPECT_HD inline int32_t run_end(const int32_t* p, int32_t c, int32_t ncol, int32_t v) {
    int64_t lo = c, hi = -1, step = 1;   // p[lo] == v; hi: first index known to hold a larger value
    while (hi < 0) {
        const int64_t t = lo + step;
        if (t > ncol) {
            hi = (int64_t)ncol + 1;
        } else if (p[t] == v) {
            lo = t;
            step <<= 1;
        } else {
            hi = t;
        }
    }
    while (hi - lo > 1) {
        const int64_t mid = lo + ((hi - lo) >> 1);
        if (p[mid] == v) lo = mid; else hi = mid;
    }
    return (int32_t)lo;
}
// stride of the lean image in dwords for a widest chunk of `widest` columns: 16-bit offsets, two per dword,
// one closing offset and a pad, whole 16-byte pieces
RSP_INSPECT_HD inline int32_t lean_stride_dwords(int32_t widest) { return ((widest + 2 + 1) / 2 + 3) & ~3; }

struct Stats {              // what the choice of form rests on; every word a flag or a maximum from zero
    int32_t invalid;        // p[] is not what a dgCMatrix guarantees (p[0] = 0, non-decreasing, p[ncol] = nnz)
    int32_t max_skip;       // snapped form: largest distance from a chunk's grid position to its first column start
    int32_t max_len;        // longest column
    int32_t inv_min_len;    // INT_MAX - the shortest column
    int32_t lean_bad;       // lean form: a column reaches more than a row past its chunk, or a chunk has more columns than the image has room for
    int32_t lean_widest;    // lean form: most columns starting in one chunk
    int32_t ready;          // device-made plans: kStatsReady once every word above has landed (written last, behind a system-scope fence)
    int32_t pad;
};
constexpr int32_t kStatsReady = 0x5253504b;
constexpr int kSpanWrites = 4;   // chunks one column start writes at most (more means a column longer than a chunk: that plan is never used)

// What the lean kernel can hold (colsums_kernels.h): elements per row of x, longest column, most columns per chunk.
struct LeanLimits {
    int32_t row_elems, max_column, max_columns;
};

// The inspection: for every chunk of the grid, the first column start at or after its grid position and the
// LAST column starting there (empty columns at that position end where the previous chunk ends: they are its).
// Pure integer work on the host copy of p[]: nchunks x 2 binary searches.
inline void inspect_offsets(const int32_t* p, int32_t ncol, int64_t nnz, const Grid& grid, std::vector<Rec>* rec,
                            int32_t* max_skip) {
    rec->resize((size_t)grid.nchunks + 1);
    int32_t worst = 0;
    const int32_t* pend = p + (size_t)ncol + 1;
    for (int32_t w = 0; w < grid.nchunks; ++w) {
        const int64_t cs = grid.start(w);
        const int32_t* first = std::lower_bound(p, pend, (int32_t)cs);          // p[ncol] = nnz > cs: always found
        const int32_t xs0 = *first;
        const int32_t* past = std::upper_bound(first, pend, xs0);
        (*rec)[w] = Rec{(int32_t)(past - p) - 1, xs0};
        const int64_t skip = (int64_t)xs0 - cs;
        if (skip > worst) worst = (int32_t)(skip > INT32_MAX ? INT32_MAX : skip);
    }
    (*rec)[grid.nchunks] = Rec{ncol, (int32_t)nnz};
    *max_skip = worst;
}
// Runs fn(begin, end) over [0, n) on a few host threads when the range is long (a plan for 1e7 columns is ~50 ms of
// integer work on one thread); anything that goes wrong with the threads falls back to the calling thread.
template <class Fn>
inline void inspect_parallel(int64_t n, int64_t grain, Fn&& fn) {
    const unsigned hw = std::thread::hardware_concurrency();
    int64_t parts = n / (grain > 0 ? grain : 1);
    if (parts > 8) parts = 8;
    if (hw > 0 && parts > (int64_t)hw) parts = hw;
    if (parts <= 1) {
        fn((int64_t)0, n);
        return;
    }
    auto range = [&](int64_t k) { fn(n * k / parts, n * (k + 1) / parts); };   // (fn only writes its own range's outputs)
    std::vector<std::thread> pool;
    int64_t started = 0;   // ranges 1..started run on threads of their own
    try {
        pool.reserve((size_t)parts - 1);
        for (int64_t k = 1; k < parts; ++k) {
            pool.emplace_back([&range, k] { range(k); });
            started = k;
        }
    } catch (...) {   // out of threads or memory: the rest runs here
    }
    range(0);
    for (int64_t k = started + 1; k < parts; ++k) range(k);
    for (auto& t : pool) t.join();
}

// The lean form's inspection (colsums_lean_kernel): applies when no column is longer than kLeanMaxColumn entries,
// no chunk (2..16 rows of x, lean_rows_setting) holds more than kLeanMaxColumns column starts, and no column reaches
// more than one row past its chunk's grid end.  Chunk w owns the columns that START in its grid range
// [cs_w, cs_{w+1}) (the last chunk: all that remain); their starts relative to cs_w fit 16 bits.  The result is ONE
// host buffer: nchunks headers {first column, columns} followed by the 16-bit offsets at a fixed stride.
inline bool inspect_lean(const int32_t* p, int32_t ncol, int64_t nnz, int32_t rows, const LeanLimits& lim,
                         std::vector<uint32_t>* image, int32_t* nchunks_out, int32_t* stride_dwords,
                         int32_t* max_columns) {
    const int64_t chunk = (int64_t)rows * lim.row_elems;
    const int64_t nchunks = (nnz + chunk - 1) / chunk;
    if (nchunks <= 0 || nchunks > INT32_MAX / 4) return false;
    std::atomic<int> too_long{0};
    inspect_parallel(ncol, 1 << 20, [&](int64_t c0, int64_t c1) {
        int bad = 0;
        for (int64_t c = c0; c < c1; ++c) bad |= (p[c + 1] - p[c] > lim.max_column);
        if (bad) too_long.store(1, std::memory_order_relaxed);
    });
    if (too_long.load()) return false;
    // first column starting at or after every chunk's grid position (chunk nchunks: ncol)
    std::vector<int32_t> first((size_t)nchunks + 1);
    const int32_t* pend = p + (size_t)ncol + 1;
    inspect_parallel(nchunks, 1 << 14, [&](int64_t w0, int64_t w1) {
        const int32_t* at = p;
        for (int64_t w = w0; w < w1; ++w) {
            at = std::lower_bound(at, pend, (int32_t)(w * chunk));   // (chunk starts ascend: search on from the last hit)
            int64_t c = at - p;
            first[(size_t)w] = (int32_t)(c > ncol ? ncol : c);
        }
    });
    first[(size_t)nchunks] = ncol;
    int32_t widest = 0;
    for (int64_t w = 0; w < nchunks; ++w) {
        const int32_t c0 = first[(size_t)w], c1 = first[(size_t)w + 1];
        if (c1 - c0 > widest) widest = c1 - c0;
        // the last owned column ends at p[c1]; the chunk has its own rows and one more
        if (c1 > c0 && (int64_t)p[c1] - w * chunk > chunk + lim.row_elems) return false;
    }
    if (widest > lim.max_columns) return false;
    const int32_t stride = lean_stride_dwords(widest);
    image->assign((size_t)nchunks * 2 + (size_t)nchunks * (size_t)stride, 0u);
    Rec* hdr = (Rec*)image->data();
    uint32_t* offs = image->data() + (size_t)nchunks * 2;
    inspect_parallel(nchunks, 1 << 13, [&](int64_t w0, int64_t w1) {
        for (int64_t w = w0; w < w1; ++w) {
            const int32_t c0 = first[(size_t)w], n = first[(size_t)w + 1] - c0;
            hdr[w] = Rec{c0, n};
            uint16_t* o = (uint16_t*)(offs + (size_t)w * (size_t)stride);
            const int64_t cs = w * chunk;
            if (n > 0)
                for (int32_t j = 0; j <= n; ++j) o[j] = (uint16_t)((int64_t)p[c0 + j] - cs);
        }
    });
    *nchunks_out = (int32_t)nchunks;
    *stride_dwords = stride;
    *max_columns = widest;
    return true;
}

// The one-pass formulation executed on the host, index by index, exactly as inspect_device.hip's three kernels do
// it: the CPU restatement of the device inspector (tests/c/inspect_selftest.cpp checks it against the two
// search-based inspectors above, also under the sanitizers; on the GPU the device images are compared with theirs).
// rec: nchunks + 1 records; lean_rows == 0: no lean part.  image is written at the stride `widest` gives.
inline void inspect_by_columns(const int32_t* p, int32_t ncol, int64_t nnz, const Grid& grid, std::vector<Rec>* rec,
                               int32_t lean_rows, const LeanLimits& lim, int32_t capacity, std::vector<uint32_t>* image,
                               int32_t* lean_chunks_out, Stats* st) {
    *st = Stats{};
    rec->assign((size_t)grid.nchunks + 1, Rec{0, 0});
    const int64_t lchunk = (int64_t)lean_rows * lim.row_elems;
    const int64_t lean_chunks = lean_rows > 0 ? (nnz + lchunk - 1) / lchunk : 0;
    const Grid lgrid{(int32_t)lchunk, (int32_t)lean_chunks, (int32_t)lchunk, (int32_t)lean_chunks};
    std::vector<int32_t> first((size_t)lean_chunks + 1, 0);
    // K1: index c in [0, ncol]
    for (int64_t c = 0; c <= ncol; ++c) {
        const int32_t v = p[c], prev = c > 0 ? p[c - 1] : -1, next = c < ncol ? p[c + 1] : INT_MAX;
        const bool bad = (c == 0 && v != 0) || v < prev || (c == ncol && v != nnz) || v < 0 || v > nnz;
        if (bad) st->invalid = 1;
        if (c < ncol && !bad && next >= v) {
            st->max_len = std::max(st->max_len, next - v);
            st->inv_min_len = std::max(st->inv_min_len, INT_MAX - (next - v));
        }
        if (!bad && (c == 0 || prev < v)) {
            const Span s = span_of(grid, prev, v);
            if (s.lo <= s.hi) {
                const int64_t sk = (int64_t)v - grid.start(s.lo);
                st->max_skip = std::max(st->max_skip, sk > INT_MAX ? INT_MAX : (int32_t)sk);
                const int32_t last = next == v ? run_end(p, (int32_t)c, ncol, v) : (int32_t)c;
                for (int32_t w = s.lo; w <= s.hi && w < s.lo + kSpanWrites; ++w) (*rec)[(size_t)w] = Rec{last, v};
            }
            if (lean_rows > 0) {
                const Span l = span_of(lgrid, prev, v);
                for (int32_t w = l.lo; w <= l.hi && w < l.lo + kSpanWrites; ++w) first[(size_t)w] = (int32_t)c;
            }
        }
    }
    (*rec)[(size_t)grid.nchunks] = Rec{ncol, (int32_t)nnz};
    *lean_chunks_out = (int32_t)lean_chunks;
    if (lean_rows <= 0) return;
    first[(size_t)lean_chunks] = ncol;
    auto clamped = [&](int64_t w, int32_t* c0, int32_t* c1) {
        *c0 = std::min(std::max(first[(size_t)w], 0), ncol);
        *c1 = std::min(std::max(first[(size_t)w + 1], *c0), ncol);
    };
    // K2: columns per chunk, reach
    for (int64_t w = 0; w < lean_chunks; ++w) {
        int32_t c0, c1;
        clamped(w, &c0, &c1);
        const int32_t n = c1 - c0;
        if (n > 0 && (int64_t)p[c1] - w * lchunk > lchunk + lim.row_elems) st->lean_bad = 1;
        if (n > capacity) st->lean_bad = 1;
        st->lean_widest = std::max(st->lean_widest, n);
    }
    // K3: the image
    const int32_t stride = std::min(lean_stride_dwords(st->lean_widest), lean_stride_dwords(capacity));
    image->assign((size_t)lean_chunks * 2 + (size_t)lean_chunks * (size_t)stride, 0u);
    Rec* hdr = (Rec*)image->data();
    uint32_t* offs = image->data() + (size_t)lean_chunks * 2;
    for (int64_t w = 0; w < lean_chunks; ++w) {
        int32_t c0, c1;
        clamped(w, &c0, &c1);
        const int32_t n = c1 - c0;
        hdr[w] = Rec{c0, n};
        uint32_t* mine = offs + (size_t)w * (size_t)stride;
        for (int32_t d = 0; d < stride; ++d) {
            uint32_t lo = 0, hi = 0;
            if (n > 0) {
                if (2 * d <= n) lo = (uint32_t)(uint16_t)((int64_t)p[c0 + 2 * d] - w * lchunk);
                if (2 * d + 1 <= n) hi = (uint32_t)(uint16_t)((int64_t)p[c0 + 2 * d + 1] - w * lchunk);
            }
            mine[d] = lo | (hi << 16);
        }
    }
}

}  // namespace inspect
}  // namespace rsp
#endif
