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

namespace rsp {
namespace inspect {

struct Rec {   // two 32-bit numbers per chunk (the device sees them as int2)
    int32_t a, b;
};

// The chunk grid of a call: `nbody` chunks of `body` elements, then chunks of `tail` elements (colsums_kernels.h ChunkMap).
struct Grid {
    int32_t body, nbody, tail, nchunks;
    int64_t start(int32_t w) const {
        return w < nbody ? (int64_t)w * body : (int64_t)nbody * body + (int64_t)(w - nbody) * tail;
    }
};

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
    const int32_t stride = ((widest + 2 + 1) / 2 + 3) & ~3;   // 16-bit offsets, two per dword, whole 16-byte pieces
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

}  // namespace inspect
}  // namespace rsp
#endif
