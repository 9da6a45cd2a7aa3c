// CPU self-test of the plan inspectors (rcppsparse_amd/csrc/inspect.hpp: the code librcppsparse_hip.so ships),
// built with and without -fsanitize=address,undefined by tests/test_plan_inspector.py.  Every result is compared with
// a naive restatement written from the definitions (reference inst/include/RcppSparse.h:220-221: column c is
// [p[c], p[c+1])), and then the plan is EXECUTED on the host the way the kernels execute it, against the plain
// column loop of reference src/example.cpp:28-30.
//   snapped plan: chunk w owns elements [xs0_w, xs0_{w+1}) and columns [c0_w, c0_{w+1}), c0_w = LAST column starting at
//                 xs0_w = first column start >= the chunk's grid position.
//   lean plan:    chunk w owns the columns that START in [cs_w, cs_{w+1}); their starts relative to cs_w as 16-bit
//                 numbers; applies iff every column <= max_column, every chunk <= max_columns columns, and the last
//                 owned column ends within the chunk's rows + one.
#include "../../rcppsparse_amd/csrc/inspect.hpp"

#include <cstdio>
#include <cstdlib>
#include <random>

using namespace rsp::inspect;

static int failures = 0;
#define CHECK(cond, ...)                                   \
    do {                                                   \
        if (!(cond)) {                                     \
            if (failures < 20) { std::printf("FAIL %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); } \
            ++failures;                                    \
        }                                                  \
    } while (0)

static std::vector<int32_t> random_offsets(std::mt19937_64& g, int kind, int32_t* ncol_out) {
    std::vector<int64_t> counts;
    auto stretch = [&](int n, int lo, int hi) {
        for (int k = 0; k < n; ++k) counts.push_back(lo + (int64_t)(g() % (uint64_t)(hi - lo + 1)));
    };
    const int parts = 1 + (int)(g() % 4);
    for (int s = 0; s < parts; ++s) {
        const int n = 1 + (int)(g() % 400);
        switch (kind == 0 ? (int)(g() % 7) : kind) {
            case 1: stretch(n, 0, 3); break;           // tiny columns and empties
            case 2: stretch(n, 5, 20); break;          // the C2 regime
            case 3: stretch(n, 0, 0); break;           // a run of empty columns
            case 4: stretch(n, 60, 70); break;         // around the lean limit
            case 5: stretch(1 + n / 40, 500, 3000); break;   // long columns
            case 6: stretch(n, 1, 1); break;           // one entry each
            default: stretch(n, 0, 64); break;
        }
    }
    std::vector<int32_t> p(counts.size() + 1, 0);
    for (size_t c = 0; c < counts.size(); ++c) p[c + 1] = p[c] + (int32_t)counts[c];
    *ncol_out = (int32_t)counts.size();
    return p;
}

static void check_snapped(const std::vector<int32_t>& p, int32_t ncol, const Grid& grid) {
    const int64_t nnz = p[ncol];
    std::vector<Rec> rec;
    int32_t max_skip = -1;
    inspect_offsets(p.data(), ncol, nnz, grid, &rec, &max_skip);
    CHECK((int32_t)rec.size() == grid.nchunks + 1, "record count");
    int32_t worst = 0;
    for (int32_t w = 0; w < grid.nchunks; ++w) {
        const int64_t cs = grid.start(w);
        int32_t xs0 = (int32_t)nnz, c_last = ncol;                 // naive: scan all column starts (index ncol = the end)
        for (int32_t c = 0; c <= ncol; ++c)
            if (p[c] >= cs) { xs0 = p[c]; break; }
        for (int32_t c = 0; c <= ncol; ++c)
            if (p[c] == xs0) c_last = c;
        CHECK(rec[w].a == c_last && rec[w].b == xs0, "chunk %d: got {%d, %d}, want {%d, %d}", w, rec[w].a, rec[w].b, c_last, xs0);
        if (xs0 - cs > worst) worst = (int32_t)(xs0 - cs);
    }
    CHECK(rec[grid.nchunks].a == ncol && rec[grid.nchunks].b == (int32_t)nnz, "closing record");
    CHECK(max_skip == worst, "max_skip %d, want %d", max_skip, worst);
    // execute: every column is owned by exactly one chunk, and the owned element ranges tile [first start, nnz)
    std::vector<int> owner(ncol, 0);
    for (int32_t w = 0; w < grid.nchunks; ++w) {
        CHECK(rec[w].b <= rec[w + 1].b && rec[w].a <= rec[w + 1].a, "records ascend at %d", w);
        for (int32_t c = rec[w].a; c < rec[w + 1].a; ++c) {
            ++owner[c];
            CHECK(p[c] >= rec[w].b && p[c + 1] <= rec[w + 1].b, "column %d outside chunk %d's elements", c, w);
        }
    }
    // columns before the first chunk's first column are empty columns at position 0 (the kernel zero-fills them)
    for (int32_t c = 0; c < ncol; ++c)
        CHECK(owner[c] == 1 || (owner[c] == 0 && c < rec[0].a && p[c + 1] == 0), "column %d owned %d times", c, owner[c]);
}

static void check_lean(const std::vector<int32_t>& p, int32_t ncol, int32_t rows, const LeanLimits& lim,
                       const std::vector<double>& x) {
    const int64_t nnz = p[ncol];
    std::vector<uint32_t> image;
    int32_t nchunks = -1, stride = -1, widest = -1;
    const bool lean = inspect_lean(p.data(), ncol, nnz, rows, lim, &image, &nchunks, &stride, &widest);
    // naive applicability
    const int64_t chunk = (int64_t)rows * lim.row_elems;
    const int64_t want_chunks = (nnz + chunk - 1) / chunk;
    bool ok = want_chunks > 0;
    for (int32_t c = 0; c < ncol && ok; ++c) ok = p[c + 1] - p[c] <= lim.max_column;
    std::vector<std::vector<int32_t> > cols((size_t)(want_chunks > 0 ? want_chunks : 0));
    if (ok) {
        for (int32_t c = 0; c < ncol; ++c) {
            int64_t w = p[c] / chunk;
            if (w >= want_chunks) w = want_chunks - 1;           // columns starting at nnz (trailing empties): the last chunk's
            cols[(size_t)w].push_back(c);
        }
        for (int64_t w = 0; w < want_chunks && ok; ++w) {
            if ((int32_t)cols[(size_t)w].size() > lim.max_columns) ok = false;
            if (!cols[(size_t)w].empty() && (int64_t)p[cols[(size_t)w].back() + 1] - w * chunk > chunk + lim.row_elems) ok = false;
        }
    }
    CHECK(lean == ok, "lean applicability: got %d, want %d (ncol %d nnz %lld rows %d)", (int)lean, (int)ok, ncol, (long long)nnz, rows);
    if (!lean || !ok) return;
    CHECK(nchunks == (int32_t)want_chunks, "chunk count %d, want %lld", nchunks, (long long)want_chunks);
    CHECK((int64_t)image.size() == (int64_t)nchunks * 2 + (int64_t)nchunks * stride, "image size");
    const Rec* hdr = (const Rec*)image.data();
    const uint32_t* offs = image.data() + (size_t)nchunks * 2;
    int32_t most = 0;
    std::vector<double> out(ncol, -1.0);
    for (int32_t w = 0; w < nchunks; ++w) {
        const std::vector<int32_t>& mine = cols[(size_t)w];
        const int32_t n = (int32_t)mine.size();
        if (n > most) most = n;
        CHECK(hdr[w].b == n && (n == 0 || hdr[w].a == mine[0]), "chunk %d header {%d, %d}, want {%d, %d}", w, hdr[w].a, hdr[w].b, n ? mine[0] : -1, n);
        CHECK((n + 2 + 1) / 2 <= stride, "chunk %d: %d columns do not fit stride %d", w, n, stride);
        if (n == 0) continue;
        const uint16_t* o = (const uint16_t*)(offs + (size_t)w * stride);
        // execute the chunk the way colsums_lean_kernel does: rows + 1 rows of x from the chunk's grid position
        const int64_t cs = (int64_t)w * chunk, avail = std::min<int64_t>(nnz - cs, chunk + lim.row_elems);
        for (int32_t j = 0; j < n; ++j) {
            const int32_t lo = o[j], hi = o[j + 1];
            CHECK(lo <= hi && hi <= avail, "chunk %d column %d: offsets %d..%d of %lld staged", w, j, lo, hi, (long long)avail);
            CHECK(cs + lo == p[hdr[w].a + j] && cs + hi == p[hdr[w].a + j + 1], "chunk %d column %d: wrong offsets", w, j);
            double s = 0.0;
            for (int32_t e = lo; e < hi; ++e) s += x[(size_t)(cs + e)];
            out[(size_t)hdr[w].a + j] = s;
        }
    }
    CHECK(widest == most, "widest %d, want %d", widest, most);
    for (int32_t c = 0; c < ncol; ++c) {   // reference src/example.cpp:28-30
        double s = 0.0;
        for (int32_t e = p[c]; e < p[c + 1]; ++e) s += x[(size_t)e];
        CHECK(out[c] == s, "column %d: lean execution %g, reference loop %g", c, out[c], s);
    }
}

// The one-pass formulation (inspect_by_columns: what the device-side inspector's three kernels compute, executed index
// by index on the host) against the two search-based inspectors: the same records wherever the plan is snapped (and the
// same max_skip always), the same lean image / stride / widest wherever the lean form applies within the capacity, and
// the same verdicts otherwise; an offsets array that is no dgCMatrix's is flagged.
static void check_by_columns(const std::vector<int32_t>& p, int32_t ncol, const Grid& grid, int32_t rows,
                             const LeanLimits& lim, int32_t capacity) {
    const int64_t nnz = p[ncol];
    std::vector<Rec> want_rec, rec;
    int32_t want_skip = -1;
    inspect_offsets(p.data(), ncol, nnz, grid, &want_rec, &want_skip);
    std::vector<uint32_t> want_image, image;
    int32_t want_chunks = -1, want_stride = -1, want_widest = -1, chunks = -1;
    const bool want_lean = inspect_lean(p.data(), ncol, nnz, rows, lim, &want_image, &want_chunks, &want_stride, &want_widest);
    Stats st;
    inspect_by_columns(p.data(), ncol, nnz, grid, &rec, rows, lim, capacity, &image, &chunks, &st);
    CHECK(st.invalid == 0, "a valid offsets array was flagged invalid");
    CHECK(st.max_skip == want_skip, "one-pass max_skip %d, searches %d", st.max_skip, want_skip);
    int32_t mx = 0, mn = INT_MAX;
    for (int32_t c = 0; c < ncol; ++c) {
        mx = std::max(mx, p[c + 1] - p[c]);
        mn = std::min(mn, p[c + 1] - p[c]);
    }
    CHECK(st.max_len == mx && INT_MAX - st.inv_min_len == mn, "column lengths %d..%d, want %d..%d", INT_MAX - st.inv_min_len, st.max_len, mn, mx);
    if (want_skip <= 512) {   // a snapped plan: its records are used, so they have to be the searches' records
        CHECK(rec.size() == want_rec.size(), "record count");
        for (size_t w = 0; w < rec.size() && w < want_rec.size(); ++w)
            CHECK(rec[w].a == want_rec[w].a && rec[w].b == want_rec[w].b, "record %zu: {%d, %d}, want {%d, %d}", w, rec[w].a,
                  rec[w].b, want_rec[w].a, want_rec[w].b);
    }
    const bool lean = !st.lean_bad && st.max_len <= lim.max_column && st.lean_widest <= lim.max_columns;
    const bool within = want_lean && want_widest <= capacity;
    CHECK(lean == within, "one-pass lean verdict %d, searches %d (widest %d, capacity %d)", (int)lean, (int)want_lean, want_widest, capacity);
    if (lean && within) {
        CHECK(chunks == want_chunks && st.lean_widest == want_widest, "lean chunks %d / widest %d, want %d / %d", chunks, st.lean_widest, want_chunks, want_widest);
        CHECK(image == want_image, "lean image differs (%zu words, want %zu)", image.size(), want_image.size());
    }
}

int main(int argc, char** argv) {
    const int cases = argc > 1 ? std::atoi(argv[1]) : 1500;
    std::mt19937_64 g(12345);
    const LeanLimits lim{128, 64, 1278};
    for (int k = 0; k < cases; ++k) {
        int32_t ncol = 0;
        std::vector<int32_t> p = random_offsets(g, k % 9 < 7 ? k % 9 : 0, &ncol);
        const int64_t nnz = p[ncol];
        if (nnz == 0) continue;                                 // (the library does not plan empty matrices)
        std::vector<double> x((size_t)nnz);
        for (auto& v : x) v = (double)((int64_t)(g() % 2001) - 1000) / 64.0;
        // chunk grids: uniform, and body + shorter tail chunks (the taper of long calls)
        const int32_t body = 128 * (1 + (int32_t)(g() % 6));
        const int32_t total_rows = (int32_t)((nnz + 127) / 128);
        Grid grid;
        if (g() % 2) {
            grid = Grid{body, (int32_t)((nnz + body - 1) / body), body, (int32_t)((nnz + body - 1) / body)};
        } else {
            const int32_t tail = 128, brows = body / 128;
            const int32_t nbody = (total_rows * 7 / 10) / brows;
            const int32_t ntail = total_rows - nbody * brows;
            grid = Grid{body, nbody, tail, nbody + ntail};
        }
        check_snapped(p, ncol, grid);
        static const int rows_choices[] = {2, 3, 4, 5, 6, 8, 12, 16};
        check_lean(p, ncol, rows_choices[g() % 8], lim, x);
        {
            static const int caps[] = {126, 200, 1278, 8, 40};
            check_by_columns(p, ncol, grid, rows_choices[g() % 8], lim, caps[g() % 5]);
            // the same array made invalid in one place: the one-pass inspector has to notice
            if (ncol >= 3 && k % 7 == 0) {
                std::vector<int32_t> q = p;
                const int kind = (int)(g() % 3);
                if (kind == 0) q[0] = 1;
                else if (kind == 1) q[ncol] = q[ncol] + 1;
                else { const size_t at = 1 + (size_t)(g() % (uint64_t)(ncol - 1)); q[at] = q[at + 1] + 1 + (int32_t)(g() % 5); }
                std::vector<Rec> r2;
                std::vector<uint32_t> im2;
                int32_t ch2 = 0;
                Stats st2;
                inspect_by_columns(q.data(), ncol, nnz, grid, &r2, 4, lim, 126, &im2, &ch2, &st2);
                CHECK(st2.invalid == 1, "an invalid offsets array (kind %d) was not flagged", kind);
            }
        }
        // a chunk holding exactly max_columns column starts, and one more
        if (k % 50 == 0) {
            for (int extra = 0; extra < 2; ++extra) {
                std::vector<int32_t> q(1, 0);
                for (int c = 0; c < 300; ++c) q.push_back(q.back() + 3);
                for (int c = 0; c < lim.max_columns + extra - 10; ++c) q.push_back(q.back());   // empties at one position
                for (int c = 0; c < 300; ++c) q.push_back(q.back() + 3);
                std::vector<double> y((size_t)q.back(), 0.5);
                check_lean(q, (int32_t)q.size() - 1, 8, lim, y);
                const int32_t qn = (int32_t)q.size() - 1;
                const Grid qg{1024, (q.back() + 1023) / 1024, 1024, (q.back() + 1023) / 1024};
                check_by_columns(q, qn, qg, 8, lim, lim.max_columns);
            }
        }
    }
    if (cases >= 2000) {   // long enough for the inspectors' worker threads (ranges of 2^20 columns, 2^14 / 2^13 chunks)
        std::vector<int32_t> q(1, 0);
        for (int c = 0; c < 2300000; ++c) q.push_back(q.back() + (int32_t)(g() % 21));
        std::vector<double> y((size_t)q.back());
        for (auto& v : y) v = (double)((int64_t)(g() % 2001) - 1000) / 64.0;
        check_lean(q, (int32_t)q.size() - 1, 4, lim, y);
    }
    if (failures) {
        std::printf("inspect selftest FAILED: %d checks\n", failures);
        return 1;
    }
    std::printf("inspect selftest ok (%d random offset arrays)\n", cases);
    return 0;
}
