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
};

namespace Rcpp {

inline SEXP mock_new() { return new SEXPREC(); }   // the test process is short-lived: no GC

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

private:
    NumericVector d_;
    int nr_, nc_;
};

inline SEXP wrap(const NumericVector& v) { SEXP s = mock_new(); s->num = v.storage(); return s; }
inline SEXP wrap(const IntegerVector& v) { SEXP s = mock_new(); s->integer = v.storage(); return s; }
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
#ifndef NULL
#define NULL 0
#endif
#endif
