#!/bin/bash
# Copies the canonical headers into inst/include so the directory is a self-contained
# R package source tree:  bash assemble.sh && R CMD INSTALL .
set -e
here=$(cd "$(dirname "$0")" && pwd)
cp "$here/../RcppSparse.h" "$here/../rcppsparse_core.hpp" "$here/../columnsums_impl.hpp" "$here/inst/include/"
cp "$here/../../../include/rcppsparse_hip.h" "$here/inst/include/"
sed -i 's#"../../include/rcppsparse_hip.h"#"rcppsparse_hip.h"#' "$here/inst/include/columnsums_impl.hpp"
# the package namespace: native routines are registered (see src/rcpp_glue.cpp), the only
# exported R function is columnSums, Rcpp and Matrix are imported like in the reference
printf '%s\n' 'useDynLib(RcppSparse, .registration=TRUE)' 'import(Rcpp)' 'import(Matrix)' \
    'export(columnSums)' 'export(columnSumsBackend)' 'export(gpuMatrix)' 'export(gpuFree)' 'export(gpuColMeans)' 'export(gpuRowSums)' \
    'export(gpuRowMeans)' 'export(gpuCrossprod)' 'S3method(dim, gpuMatrix)' 'S3method(dim, gpuMatrixMulti)' > "$here/NAMESPACE"
echo "assembled: $(ls $here/inst/include | tr '\n' ' ')"
