// Host-mirror self-test under AddressSanitizer + UBSan: drives the Rcpp-free seam's CPU entry points
// (iterators, element access, reductions, transpose, crossprod, restricted iterators, row iterator)
// over random small matrices, including empty columns / rows and empty row sets -- the corners where
// the reference's own restricted iterators read out of bounds (RcppSparse.h:242, :299).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

extern "C" {
int seam_columnSums_by_iterator(const double*, const int*, const int*, const int*, int, double*);
int seam_sizes(const double*, const int*, const int*, const int*, int, int, unsigned int*);
int seam_walk_column(const double*, const int*, const int*, const int*, int, int, int*, double*, int*, int*);
int seam_walk_restricted(const double*, const int*, const int*, const int*, int, int, const unsigned int*, int, int,
                         int*, double*, int*);
int seam_dense(const double*, const int*, const int*, const int*, int, int, int, int, double*);
int seam_transpose(const double*, const int*, const int*, const int*, int, double*, int*, int*);
int seam_is_appx_symmetric(const double*, const int*, const int*, const int*, int, int*);
int seam_construct_from_s4(int);
int seam_shares_storage(void);
}

static uint64_t rng = 0x9E3779B97F4A7C15ull;
static uint32_t rnd(uint32_t n) { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (uint32_t)(rng % n); }

int main() {
    for (int iter = 0; iter < 1500; ++iter) {
        const int nrow = 1 + (int)rnd(40), ncol = 1 + (int)rnd(30);
        std::vector<int> p(ncol + 1, 0), ri;
        std::vector<double> x;
        for (int c = 0; c < ncol; ++c) {
            for (int r = 0; r < nrow; ++r)
                if (rnd(5) == 0) { ri.push_back(r); x.push_back((double)rnd(2000) / 100.0 - 10.0); }
            p[c + 1] = (int)x.size();
        }
        const int nnz = (int)x.size();
        const int dim[2] = {nrow, ncol};
        x.push_back(0.0); ri.push_back(0);   // keep data() non-null for empty matrices
        std::vector<double> out(std::max(nrow, ncol) * std::max(nrow, ncol) + 4);
        std::vector<int> idx(std::max(nrow, ncol) + 4), cols(nrow + 4);
        std::vector<double> vals(std::max(nrow, ncol) + 4);
        unsigned int sz[6];
        int n = 0, flag = 0;
        if (seam_columnSums_by_iterator(x.data(), ri.data(), p.data(), dim, nnz, out.data())) return 1;
        if (seam_sizes(x.data(), ri.data(), p.data(), dim, nnz, (int)rnd(ncol), sz)) return 2;
        const int c = (int)rnd(ncol), r = (int)rnd(nrow);
        if (seam_walk_column(x.data(), ri.data(), p.data(), dim, nnz, c, idx.data(), vals.data(), cols.data(), &n)) return 3;
        std::vector<unsigned int> s;
        for (int q = 0; q < nrow; ++q) if (rnd(3) == 0) s.push_back((unsigned)q);
        for (int mode = 0; mode < 2; ++mode) {
            if (seam_walk_restricted(x.data(), ri.data(), p.data(), dim, nnz, c, s.data(), (int)s.size(), mode,
                                     idx.data(), vals.data(), &n)) return 4;
            if (seam_walk_restricted(x.data(), ri.data(), p.data(), dim, nnz, c, s.data(), 0, mode, idx.data(),
                                     vals.data(), &n)) return 5;          // empty row set
        }
        if (seam_walk_restricted(x.data(), ri.data(), p.data(), dim, nnz, r, s.data(), 0, 2, idx.data(), vals.data(), &n)) return 6;
        for (int which = 0; which <= 7; ++which)
            if (seam_dense(x.data(), ri.data(), p.data(), dim, nnz, which, which == 4 ? c : r, c, out.data())) return 7;
        std::vector<double> tx(nnz + 1);
        std::vector<int> ti(nnz + 1), tp(nrow + 1);
        if (seam_transpose(x.data(), ri.data(), p.data(), dim, nnz, tx.data(), ti.data(), tp.data())) return 8;
        if (seam_is_appx_symmetric(x.data(), ri.data(), p.data(), dim, nnz, &flag)) return 9;
    }
    if (seam_construct_from_s4(0xF) != 0 || seam_construct_from_s4(0x7) != 1 || seam_shares_storage() != 1) return 10;
    std::printf("host selftest ok\n");
    return 0;
}
