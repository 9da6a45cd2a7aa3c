// Compiles THIS repository's Rcpp layer (drop-in header + exported columnSums + glue) against the
// API-shaped mock in this directory and drives it the way R would: package init, routine lookup
// in the registration table, .Call with a dgCMatrix-like S4 object.
//   driver registered          -> prints the registered routine name and arity
//   driver missing_slot        -> S4 without `p`: the glue must turn the exception into an R error
//   driver kat                 -> Documentation.Rmd:213-216 matrix through .Call (needs a GPU)
//   driver handle              -> gpuMatrix(A) external pointer: resident sums, copy semantics, finalizer (GPU)
//   driver handle_nogpu        -> gpuMatrix(A) on a machine without a GPU must be an R error
//   driver kat_cpu             -> the same matrix on a machine WITHOUT a GPU: the host loop answers (reference bits),
//                                 columnSumsBackend() says "cpu"; with options(RcppSparse.require_gpu = TRUE) an R error
//   driver min_nnz             -> the offload threshold on a machine WITH a GPU: options(RcppSparse.min_nnz = n) /
//                                 RCPPSPARSE_MIN_NNZ decide between the host loop and the device; a required GPU overrides
//   driver backend             -> prints columnSumsBackend() / columnSumsBackend(last = TRUE) before any call
//   driver handle_swap         -> a "gpuMatrixMulti" handed to a single-GPU routine (and the reverse), an edited Dim (GPU)
//   driver handle_methods      -> colMeans / rowSums / rowMeans / crossprod on the handle, and the same matrix
//                                 spread over three shards (gpuMatrix(A, devices = c(0, 0, 0))) (GPU)
#include "../../rcppsparse_amd/host/RcppSparse.h"

#include <cstdio>
#include <cstring>

extern "C" SEXP _RcppSparse_columnSums(SEXP);
extern "C" void R_init_RcppSparse(DllInfo*);
extern "C" void R_unload_RcppSparse(DllInfo*);

static SEXP dgc(bool with_p) {
    Rcpp::S4 A(std::string("dgCMatrix"));
    const double x[5] = {0.41, 0.35, 0.84, 0.37, 0.26};
    const int i[5] = {0, 2, 0, 1, 1}, p[6] = {0, 0, 1, 2, 4, 5}, dim[2] = {5, 5};
    A.slot("x") = Rcpp::NumericVector(x, x + 5);
    A.slot("i") = Rcpp::IntegerVector(i, i + 5);
    if (with_p) A.slot("p") = Rcpp::IntegerVector(p, p + 6);
    A.slot("Dim") = Rcpp::IntegerVector(dim, dim + 2);
    return A;
}

int main(int argc, char** argv) {
    const std::string mode = argc > 1 ? argv[1] : "registered";
    DllInfo dll = {0, true};
    R_init_RcppSparse(&dll);
    if (!dll.registered || dll.dynamic_symbols) return 10;
    if (mode == "registered") {
        for (int k = 0; dll.registered[k].name; ++k)
            std::printf("%s %d\n", dll.registered[k].name, dll.registered[k].numArgs);
        return 0;
    }
    typedef SEXP (*call1)(SEXP);
    typedef SEXP (*call2)(SEXP, SEXP);
    call1 fn = (call1)dll.registered[0].fun;          // what .Call(`_RcppSparse_columnSums`, A) resolves to
    if (mode == "handle" || mode == "handle_nogpu") {
        call2 gpu_matrix = 0;
        call1 gpu_sums = 0, gpu_free = 0;
        for (int k = 0; dll.registered[k].name; ++k) {
            const std::string nm = dll.registered[k].name;
            if (nm == "_RcppSparse_gpuMatrix") gpu_matrix = (call2)dll.registered[k].fun;
            if (nm == "_RcppSparse_gpuColumnSums") gpu_sums = (call1)dll.registered[k].fun;
            if (nm == "_RcppSparse_gpuFree") gpu_free = (call1)dll.registered[k].fun;
        }
        if (!gpu_matrix || !gpu_sums || !gpu_free) return 20;
        SEXP A = dgc(true);
        SEXP dev = Rcpp::wrap(Rcpp::IntegerVector::create(0, 0));
        SEXP h = gpu_matrix(A, dev);
        if (mode == "handle_nogpu") {
            std::printf("%s\n", h->error.c_str());
            return h->error.empty() ? 21 : 0;
        }
        if (!h->error.empty()) { std::printf("R error: %s\n", h->error.c_str()); return 22; }
        if (!h->is_extptr || !h->extptr || h->attrs.at("class")->str != "gpuMatrix") return 23;
        const double want[5] = {0.0, 0.41, 0.35, 0.84 + 0.37, 0.26};
        SEXP r = gpu_sums(h);
        if (!r->error.empty() || r->num->size() != 5 || std::memcmp(&(*r->num)[0], want, sizeof want) != 0) return 24;
        // the handle is a COPY: changing the R object in place (Documentation.Rmd:335-347) is not seen
        (*A->slots.at("x")->num)[0] = 99.0;
        r = gpu_sums(h);
        if (std::memcmp(&(*r->num)[0], want, sizeof want) != 0) return 25;
        SEXP fresh = fn(A);                             // ... while the one-shot path sees the new value
        if ((*fresh->num)[1] != 99.0) return 26;
        // a second handle, dropped without gpuFree: R's collector runs the finalizer exactly once
        SEXP h2 = gpu_matrix(A, dev);
        if (!h2->error.empty()) return 27;
        Rcpp::mock_collect(h2);
        if (h2->extptr != 0) return 28;
        Rcpp::mock_collect(h2);                         // (already cleared: nothing happens)
        // explicit release, then use: an R error, not a crash
        if (!gpu_free(h)->error.empty() || h->extptr != 0) return 29;
        r = gpu_sums(h);
        if (r->error.find("released") == std::string::npos) return 30;
        if (!gpu_free(h)->error.empty()) return 31;     // releasing twice is harmless
        if (Rcpp::RNGScope::live() != 0 || Rcpp::RNGScope::entered() < 6) return 32;
        std::printf("gpuMatrix handle ok\n");
        return 0;
    }
    if (mode == "handle_methods") {
        // R: h <- gpuMatrix(A); gpuColMeans(h); gpuRowSums(h); gpuRowMeans(h); gpuCrossprod(h)
        //    m <- gpuMatrix(A, devices = c(0, 0, 0)); columnSums(m); gpuColMeans(m); gpuRowSums(m); gpuRowMeans(m)
        call2 gpu_matrix = 0, gpu_reduce = 0, gpu_multi = 0, multi_reduce = 0;
        call1 gpu_cross = 0, gpu_free = 0, multi_free = 0;
        for (int k = 0; dll.registered[k].name; ++k) {
            const std::string nm = dll.registered[k].name;
            if (nm == "_RcppSparse_gpuMatrix") gpu_matrix = (call2)dll.registered[k].fun;
            if (nm == "_RcppSparse_gpuReduce") gpu_reduce = (call2)dll.registered[k].fun;
            if (nm == "_RcppSparse_gpuCrossprod") gpu_cross = (call1)dll.registered[k].fun;
            if (nm == "_RcppSparse_gpuFree") gpu_free = (call1)dll.registered[k].fun;
            if (nm == "_RcppSparse_gpuMatrixMulti") gpu_multi = (call2)dll.registered[k].fun;
            if (nm == "_RcppSparse_gpuMultiReduce") multi_reduce = (call2)dll.registered[k].fun;
            if (nm == "_RcppSparse_gpuFreeMulti") multi_free = (call1)dll.registered[k].fun;
        }
        if (!gpu_matrix || !gpu_reduce || !gpu_cross || !gpu_free || !gpu_multi || !multi_reduce || !multi_free) return 40;
        // the matrix of Documentation.Rmd:213-216: columns {}, {(0, .41)}, {(2, .35)}, {(0, .84), (1, .37)}, {(1, .26)}
        const double cs[5] = {0.0, 0.41, 0.35, 0.84 + 0.37, 0.26};
        const double rs[5] = {0.41 + 0.84, 0.37 + 0.26, 0.35, 0.0, 0.0};
        double cm[5], rm[5], xp[25] = {0};
        for (int k = 0; k < 5; ++k) { cm[k] = cs[k] / 5; rm[k] = rs[k] / 5; }
        xp[1 * 5 + 1] = 0.41 * 0.41;
        xp[2 * 5 + 2] = 0.35 * 0.35;
        xp[3 * 5 + 3] = 0.84 * 0.84 + 0.37 * 0.37;
        xp[4 * 5 + 4] = 0.26 * 0.26;
        xp[1 * 5 + 3] = xp[3 * 5 + 1] = 0.41 * 0.84;          // columns 1 and 3 share row 0
        xp[3 * 5 + 4] = xp[4 * 5 + 3] = 0.37 * 0.26;          // columns 3 and 4 share row 1
        const double* want[4] = {cs, cm, rs, rm};
        SEXP A = dgc(true);
        SEXP h = gpu_matrix(A, Rcpp::wrap(Rcpp::IntegerVector::create(0, 0)));
        if (!h->error.empty()) { std::printf("R error: %s\n", h->error.c_str()); return 41; }
        for (int what = 0; what < 4; ++what) {
            SEXP r = gpu_reduce(h, Rcpp::wrap(Rcpp::IntegerVector::create(what, 0)));
            if (!r->error.empty()) { std::printf("R error: %s\n", r->error.c_str()); return 42; }
            if (r->num->size() != 5 || std::memcmp(&(*r->num)[0], want[what], 5 * sizeof(double)) != 0) return 43 + what;
        }
        if (gpu_reduce(h, Rcpp::wrap(Rcpp::IntegerVector::create(7, 0)))->error.find("what must be") == std::string::npos) return 47;
        SEXP c = gpu_cross(h);
        if (!c->error.empty()) { std::printf("R error: %s\n", c->error.c_str()); return 48; }
        if (c->num->size() != 25 || std::memcmp(&(*c->num)[0], xp, sizeof xp) != 0) return 49;
        if ((*c->attrs.at("dim")->integer)[0] != 5 || (*c->attrs.at("dim")->integer)[1] != 5) return 50;
        if (!gpu_free(h)->error.empty()) return 51;
        if (gpu_reduce(h, Rcpp::wrap(Rcpp::IntegerVector::create(2, 0)))->error.find("released") == std::string::npos) return 52;
        if (gpu_cross(h)->error.find("released") == std::string::npos) return 53;
        // three shards (all on device 0 here): same answers, bit for bit on this matrix
        const int three[3] = {0, 0, 0};
        SEXP m = gpu_multi(A, Rcpp::wrap(Rcpp::IntegerVector(three, three + 3)));
        if (!m->error.empty()) { std::printf("R error: %s\n", m->error.c_str()); return 54; }
        if (!m->is_extptr || m->attrs.at("class")->str != "gpuMatrixMulti") return 55;
        for (int what = 0; what < 4; ++what) {
            SEXP r = multi_reduce(m, Rcpp::wrap(Rcpp::IntegerVector::create(what, 0)));
            if (!r->error.empty()) { std::printf("R error: %s\n", r->error.c_str()); return 56; }
            if (r->num->size() != 5 || std::memcmp(&(*r->num)[0], want[what], 5 * sizeof(double)) != 0) return 57 + what;
        }
        Rcpp::mock_collect(m);                          // the collector's finalizer frees every shard
        if (m->extptr != 0) return 61;
        if (multi_reduce(m, Rcpp::wrap(Rcpp::IntegerVector::create(0, 0)))->error.find("released") == std::string::npos) return 62;
        if (!multi_free(m)->error.empty()) return 63;
        if (Rcpp::RNGScope::live() != 0) return 64;
        std::printf("gpuMatrix methods ok\n");
        return 0;
    }
    if (mode == "kat_cpu" || mode == "backend") {
        call1 backend = 0;
        for (int k = 0; dll.registered[k].name; ++k)
            if (std::string(dll.registered[k].name) == "_RcppSparse_columnSumsBackend") backend = (call1)dll.registered[k].fun;
        if (!backend) return 70;
        SEXP now = Rcpp::wrap(Rcpp::IntegerVector::create(0, 0)), last = Rcpp::wrap(Rcpp::IntegerVector::create(1, 0));
        if (mode == "backend") {
            std::printf("%s %s\n", backend(now)->str.c_str(), backend(last)->str.c_str());
            return 0;
        }
        if (backend(last)->str != "none") return 71;              // nothing has been summed yet
        SEXP r = fn(dgc(true));
        if (!r->error.empty()) { std::printf("R error: %s\n", r->error.c_str()); return 72; }
        const double want[5] = {0.0, 0.41, 0.35, 0.84 + 0.37, 0.26};
        if (r->num->size() != 5 || std::memcmp(&(*r->num)[0], want, sizeof want) != 0) return 73;
        if (!r->attrs.empty() || !r->klass.empty()) return 74;    // a plain numeric vector, no attributes
        for (int k = 0; k < 5; ++k) std::printf("%a%c", (*r->num)[k], k == 4 ? '\n' : ' ');
        std::printf("backend now=%s last=%s\n", backend(now)->str.c_str(), backend(last)->str.c_str());
        // options(RcppSparse.require_gpu = TRUE): the same call is an R error, and it says why
        Rcpp::mock_options()["RcppSparse.require_gpu"] = Rcpp::wrap(Rcpp::IntegerVector::create(1, 0));
        SEXP e = fn(dgc(true));
        std::printf("required: %s | now=%s\n", e->error.c_str(), backend(now)->str.c_str());
        if (e->error.find("no HIP device") == std::string::npos) return 75;
        // options(RcppSparse.require_gpu = FALSE) wins over the environment variable
        Rcpp::mock_options()["RcppSparse.require_gpu"] = Rcpp::wrap(Rcpp::IntegerVector::create(0, 0));
        if (!fn(dgc(true))->error.empty()) return 76;
        return 0;
    }
    if (mode == "min_nnz") {
        call1 backend = 0;
        for (int k = 0; dll.registered[k].name; ++k)
            if (std::string(dll.registered[k].name) == "_RcppSparse_columnSumsBackend") backend = (call1)dll.registered[k].fun;
        if (!backend) return 90;
        SEXP last = Rcpp::wrap(Rcpp::IntegerVector::create(1, 0));
        const double want[5] = {0.0, 0.41, 0.35, 0.84 + 0.37, 0.26};
        // the vignette's matrix holds 5 stored entries: one line per setting -- "<setting> <backend that answered>"
        struct { const char* name; int min_nnz; int require; } cases[] = {
            {"default", -2, -1}, {"min_nnz=0", 0, -1}, {"min_nnz=6", 6, -1}, {"min_nnz=5", 5, -1}, {"min_nnz=6+require_gpu", 6, 1}};
        for (unsigned k = 0; k < sizeof cases / sizeof cases[0]; ++k) {
            Rcpp::mock_options().erase("RcppSparse.min_nnz");
            Rcpp::mock_options().erase("RcppSparse.require_gpu");
            if (cases[k].min_nnz >= 0)
                Rcpp::mock_options()["RcppSparse.min_nnz"] = Rcpp::wrap(Rcpp::NumericVector(1, (double)cases[k].min_nnz));   // options(RcppSparse.min_nnz = 6): a double, as R stores it
            if (cases[k].require >= 0)
                Rcpp::mock_options()["RcppSparse.require_gpu"] = Rcpp::wrap(Rcpp::IntegerVector::create(cases[k].require, 0));
            SEXP r = fn(dgc(true));
            if (!r->error.empty()) { std::printf("R error: %s\n", r->error.c_str()); return 91; }
            if (r->num->size() != 5 || std::memcmp(&(*r->num)[0], want, sizeof want) != 0) return 92;   // the same bits either way
            std::printf("%s %s\n", cases[k].name, backend(last)->str.c_str());
        }
        return 0;
    }
    if (mode == "handle_swap") {
        call2 gpu_matrix = 0, gpu_reduce = 0, gpu_multi = 0, multi_reduce = 0;
        call1 gpu_sums = 0, gpu_free = 0, multi_free = 0;
        for (int k = 0; dll.registered[k].name; ++k) {
            const std::string nm = dll.registered[k].name;
            if (nm == "_RcppSparse_gpuMatrix") gpu_matrix = (call2)dll.registered[k].fun;
            if (nm == "_RcppSparse_gpuReduce") gpu_reduce = (call2)dll.registered[k].fun;
            if (nm == "_RcppSparse_gpuColumnSums") gpu_sums = (call1)dll.registered[k].fun;
            if (nm == "_RcppSparse_gpuFree") gpu_free = (call1)dll.registered[k].fun;
            if (nm == "_RcppSparse_gpuMatrixMulti") gpu_multi = (call2)dll.registered[k].fun;
            if (nm == "_RcppSparse_gpuMultiReduce") multi_reduce = (call2)dll.registered[k].fun;
            if (nm == "_RcppSparse_gpuFreeMulti") multi_free = (call1)dll.registered[k].fun;
        }
        SEXP A = dgc(true);
        SEXP h = gpu_matrix(A, Rcpp::wrap(Rcpp::IntegerVector::create(0, 0)));
        const int two[2] = {0, 0};
        SEXP m = gpu_multi(A, Rcpp::wrap(Rcpp::IntegerVector(two, two + 2)));
        if (!h->error.empty() || !m->error.empty()) return 80;
        SEXP zero = Rcpp::wrap(Rcpp::IntegerVector::create(0, 0));
        // class(m) <- "gpuMatrix"; columnSums(m): the tag, not the class, decides -- an R error, not a crash
        if (gpu_sums(m)->error.find("not a gpuMatrix handle") == std::string::npos) return 81;
        if (gpu_reduce(m, zero)->error.find("not a gpuMatrix handle") == std::string::npos) return 82;
        if (multi_reduce(h, zero)->error.find("not a gpuMatrixMulti handle") == std::string::npos) return 83;
        if (gpu_free(m)->error.find("not a gpuMatrix handle") == std::string::npos) return 84;
        if (multi_free(h)->error.find("not a gpuMatrixMulti handle") == std::string::npos) return 85;
        // attr(h, "Dim") <- c(5L, 5000L): the output length comes from the native handle
        (*h->attrs.at("Dim")->integer)[1] = 5000;
        (*m->attrs.at("Dim")->integer)[0] = 7000;
        SEXP r = gpu_sums(h);
        if (!r->error.empty() || r->num->size() != 5) return 86;
        r = multi_reduce(m, Rcpp::wrap(Rcpp::IntegerVector::create(2, 0)));
        if (!r->error.empty() || r->num->size() != 5) return 87;
        if (!gpu_free(h)->error.empty() || !multi_free(m)->error.empty()) return 88;
        std::printf("handle swap ok\n");
        return 0;
    }
    if (mode == "missing_slot") {
        SEXP r = fn(dgc(false));
        std::printf("%s\n", r->error.c_str());
        return r->error.empty() ? 12 : 0;
    }
    if (mode == "kat") {
        RcppSparse::Matrix M{Rcpp::S4(dgc(true))};
        if (M.cols() != 5 || M.n_nonzero() != 5) return 13;
        Rcpp::S4 back = M.wrap();                      // round trip shares the vectors
        if (!back.hasSlot("x") || RcppSparse::Matrix(back).x.storage() != M.x.storage()) return 14;
        RcppSparse::Matrix T = M.transpose();
        if (T.cols() != 5 || T.at(1, 3) != M.at(3, 1)) return 15;
        SEXP r = fn(dgc(true));
        if (!r->error.empty()) { std::printf("R error: %s\n", r->error.c_str()); return 16; }
        const double want[5] = {0.0, 0.41, 0.35, 0.84 + 0.37, 0.26};
        if (r->num->size() != 5 || std::memcmp(&(*r->num)[0], want, sizeof want) != 0) return 17;
        R_unload_RcppSparse(&dll);                     // the unload hook gives the library's kept buffers back ...
        r = fn(dgc(true));                             // ... and a later call simply allocates again
        if (!r->error.empty() || std::memcmp(&(*r->num)[0], want, sizeof want) != 0) return 18;
        std::printf("columnSums via .Call ok\n");
        return 0;
    }
    return 1;
}
