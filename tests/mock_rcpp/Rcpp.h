// TEST SCAFFOLDING -- see RcppCommon.h in this directory.
#ifndef MOCK_RCPP_H
#define MOCK_RCPP_H
#include "RcppCommon.h"

#include <exception>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

// An "R object": numeric vector, integer vector, S4 bag of slots, or an error marker.
struct SEXPREC {
    std::shared_ptr<std::vector<double> > num;
    std::shared_ptr<std::vector<int> > integer;
    std::map<std::string, SEXP> slots;
    std::string klass, error;
    bool is_s4 = false;
    // external pointers (Rcpp::XPtr): address, C finalizer, attributes; character scalars
    bool is_extptr = false;
    void* extptr = 0;
    void (*finalizer)(SEXP) = 0;
    std::map<std::string, SEXP> attrs;
    std::string str;
    SEXP tag = 0;          // R_ExternalPtrTag
};

namespace Rcpp {

inline SEXP mock_new() { return new SEXPREC(); }   // the test process is short-lived: no GC
inline SEXP mock_nil() { static SEXPREC nil; return &nil; }
// what R's garbage collector does to an unreachable external pointer: run its finalizer once
inline void mock_collect(SEXP s) {
    if (s && s->is_extptr && s->finalizer) {
        void (*f)(SEXP) = s->finalizer;
        s->finalizer = 0;
        f(s);
    }
}

template <class T>
class MockVector {
public:
    MockVector() : d_(std::make_shared<std::vector<T> >()) {}
    MockVector(R_xlen_t n) : d_(std::make_shared<std::vector<T> >((std::size_t)n, T())) {}
    MockVector(R_xlen_t n, T fill) : d_(std::make_shared<std::vector<T> >((std::size_t)n, fill)) {}
    template <class It>
    MockVector(It a, It b) : d_(std::make_shared<std::vector<T> >(a, b)) {}
    explicit MockVector(std::shared_ptr<std::vector<T> > d) : d_(d) {}
    T& operator[](R_xlen_t k) { return (*d_)[(std::size_t)k]; }
    const T& operator[](R_xlen_t k) const { return (*d_)[(std::size_t)k]; }
    T& operator()(R_xlen_t k) { return (*d_)[(std::size_t)k]; }
    R_xlen_t size() const { return (R_xlen_t)d_->size(); }
    typename std::vector<T>::iterator begin() { return d_->begin(); }
    typename std::vector<T>::iterator end() { return d_->end(); }
    std::shared_ptr<std::vector<T> > storage() const { return d_; }

protected:
    std::shared_ptr<std::vector<T> > d_;   // copies share storage, like Rcpp vectors
};

class NumericVector : public MockVector<double> {
public:
    using MockVector<double>::MockVector;
    NumericVector() {}
    NumericVector(SEXP s) : MockVector<double>(s->num) {}
};
class IntegerVector : public MockVector<int> {
public:
    using MockVector<int>::MockVector;
    IntegerVector() {}
    IntegerVector(SEXP s) : MockVector<int>(s->integer) {}
    static IntegerVector create(int a, int b) {
        IntegerVector v(2);
        v[0] = a;
        v[1] = b;
        return v;
    }
};
class NumericMatrix {
public:
    NumericMatrix() : nr_(0), nc_(0) {}
    NumericMatrix(int nr, int nc) : d_((R_xlen_t)nr * nc), nr_(nr), nc_(nc) {}
    double& operator()(int r, int c) { return d_[(R_xlen_t)c * nr_ + r]; }
    int nrow() const { return nr_; }
    int ncol() const { return nc_; }
    const NumericVector& data() const { return d_; }

private:
    NumericVector d_;
    int nr_, nc_;
};

inline SEXP wrap(const NumericVector& v) { SEXP s = mock_new(); s->num = v.storage(); return s; }
inline SEXP wrap(const IntegerVector& v) { SEXP s = mock_new(); s->integer = v.storage(); return s; }
inline SEXP wrap(const NumericMatrix& m) {   // a numeric vector with a dim attribute, like R's matrices
    SEXP s = mock_new();
    s->num = m.data().storage();
    s->attrs["dim"] = wrap(IntegerVector::create(m.nrow(), m.ncol()));
    return s;
}
template <class V>
inline V clone(const V& v) {
    V c(v.size());
    for (R_xlen_t k = 0; k < v.size(); ++k) c[k] = v[k];
    return c;
}

class S4 {
public:
    class SlotProxy {
    public:
        SlotProxy(SEXP owner, const std::string& name) : owner_(owner), name_(name) {}
        operator NumericVector() const { return NumericVector(owner_->slots.at(name_)); }
        operator IntegerVector() const { return IntegerVector(owner_->slots.at(name_)); }
        SlotProxy& operator=(const NumericVector& v) { owner_->slots[name_] = wrap(v); return *this; }
        SlotProxy& operator=(const IntegerVector& v) { owner_->slots[name_] = wrap(v); return *this; }

    private:
        SEXP owner_;
        std::string name_;
    };
    S4(SEXP s) : s_(s) {
        if (!s || !s->is_s4) throw std::invalid_argument("not an S4 object");
    }
    explicit S4(const std::string& klass) : s_(mock_new()) { s_->is_s4 = true; s_->klass = klass; }
    bool hasSlot(const std::string& name) const { return s_->slots.count(name) != 0; }
    SlotProxy slot(const std::string& name) const { return SlotProxy(s_, name); }
    operator SEXP() const { return s_; }

private:
    SEXP s_;
};

inline SEXP wrap(const char* text) { SEXP s = mock_new(); s->str = text; return s; }
inline SEXP wrap(SEXP s) { return s; }

template <class T>
T as(SEXP s);
template <>
inline int as<int>(SEXP s) {
    if (!s || !s->integer || s->integer->empty()) throw std::invalid_argument("Expecting a single integer value");
    return (*s->integer)[0];
}

// GetRNGstate() / PutRNGstate() bracket of the generated glue: counted so a test can see it ran
struct RNGScope {
    static int& live() { static int n = 0; return n; }
    static int& entered() { static int n = 0; return n; }
    RNGScope() { ++live(); ++entered(); }
    ~RNGScope() { --live(); }
};

template <class T>
class PreserveStorage {};
template <class T>
void standard_delete_finalizer(T* p) { delete p; }

// Rcpp::XPtr: an R external pointer owning a C object, released by a C finalizer when R
// collects it (or by release()).  Same template signature as Rcpp's.
template <class T, template <class> class StoragePolicy = PreserveStorage,
          void Finalizer(T*) = standard_delete_finalizer<T>, bool finalizeOnExit = false>
class XPtr {
public:
    class AttrProxy {
    public:
        AttrProxy(SEXP owner, const std::string& name) : owner_(owner), name_(name) {}
        AttrProxy& operator=(const char* text) { owner_->attrs[name_] = wrap(text); return *this; }
        AttrProxy& operator=(const IntegerVector& v) { owner_->attrs[name_] = wrap(v); return *this; }
        operator IntegerVector() const { return IntegerVector(owner_->attrs.at(name_)); }

    private:
        SEXP owner_;
        std::string name_;
    };
    explicit XPtr(SEXP s) : s_(s) {
        if (!s || !s->is_extptr) throw std::invalid_argument("Expecting an external pointer");
    }
    explicit XPtr(T* p, bool set_delete_finalizer = true, SEXP tag = 0, SEXP prot = 0) : s_(mock_new()) {
        (void)prot;
        s_->is_extptr = true;
        s_->extptr = p;
        s_->tag = tag;
        if (set_delete_finalizer) s_->finalizer = &finalize;
    }
    T* get() const { return static_cast<T*>(s_->extptr); }
    void release() { finalize(s_); }                 // finalizer now, pointer cleared
    AttrProxy attr(const std::string& name) const { return AttrProxy(s_, name); }
    operator SEXP() const { return s_; }

private:
    static void finalize(SEXP s) {
        T* p = static_cast<T*>(s->extptr);
        if (p) {
            s->extptr = 0;                           // R_ClearExternalPtr
            Finalizer(p);
        }
    }
    SEXP s_;
};

class RObject {
public:
    RObject() : s_(0) {}
    RObject(SEXP s) : s_(s) {}
    RObject& operator=(SEXP s) { s_ = s; return *this; }
    operator SEXP() const { return s_; }

private:
    SEXP s_;
};

namespace traits {
// what Rcpp's ReferenceInputParameter does: build the object through the Exporter and hand
// out a reference to it for the duration of the call
template <class T>
struct input_parameter;
template <class T>
struct input_parameter<T&> {
    class type {
    public:
        type(SEXP s) : obj_(Exporter<T>(s).get()) {}
        operator T&() { return obj_; }

    private:
        T obj_;
    };
};
}  // namespace traits
}  // namespace Rcpp

// C++ exception -> "R error" (here: an SEXP carrying the message)
#define BEGIN_RCPP try {
#define END_RCPP                                   \
    }                                              \
    catch (std::exception & ex_) {                 \
        SEXP err_ = Rcpp::mock_new();              \
        err_->error = ex_.what();                  \
        return err_;                               \
    }

typedef enum { FALSE = 0, TRUE } Rboolean;
typedef void* (*DL_FUNC)();
struct R_CallMethodDef { const char* name; DL_FUNC fun; int numArgs; };
struct DllInfo { const R_CallMethodDef* registered; bool dynamic_symbols; };
inline int R_registerRoutines(DllInfo* dll, const void*, const R_CallMethodDef* call, const void*, const void*) {
    dll->registered = call;
    return 1;
}
inline int R_useDynamicSymbols(DllInfo* dll, Rboolean v) { dll->dynamic_symbols = (v != FALSE); return 1; }
#define R_NilValue (Rcpp::mock_nil())
// the few C-level R entry points the package uses: symbols, options(), external pointer tags
namespace Rcpp {
inline std::map<std::string, SEXP>& mock_options() { static std::map<std::string, SEXP> o; return o; }
inline std::map<std::string, SEXP>& mock_symbols() { static std::map<std::string, SEXP> o; return o; }
}
inline SEXP Rf_install(const char* name) {          // symbols are unique per name, like R's symbol table
    SEXP& s = Rcpp::mock_symbols()[name];
    if (!s) { s = Rcpp::mock_new(); s->str = name; }
    return s;
}
inline SEXP Rf_GetOption1(SEXP sym) {
    std::map<std::string, SEXP>::const_iterator it = Rcpp::mock_options().find(sym->str);
    return it == Rcpp::mock_options().end() ? R_NilValue : it->second;
}
inline int Rf_asLogical(SEXP s) { return (s && s->integer && !s->integer->empty() && (*s->integer)[0] != 0) ? TRUE : FALSE; }
inline double Rf_asReal(SEXP s) {
    if (s && s->num && !s->num->empty()) return (*s->num)[0];
    if (s && s->integer && !s->integer->empty()) return (double)(*s->integer)[0];
    return -1.0;
}
inline SEXP R_ExternalPtrTag(SEXP s) { return (s && s->tag) ? s->tag : R_NilValue; }
#ifndef NULL
#define NULL 0
#endif
#endif
