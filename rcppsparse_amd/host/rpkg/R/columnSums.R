#' Column sums of a sparse matrix on the GPU
#'
#' @param A an object of class \code{dgCMatrix}, or a \code{"gpuMatrix"} handle made by
#'   \code{\link{gpuMatrix}} (the matrix is then already in GPU memory and is not transferred)
#' @return numeric vector of length \code{ncol(A)}
#' @export
columnSums <- function(A) {
    if (inherits(A, "gpuMatrix")) .Call(`_RcppSparse_gpuColumnSums`, A)
    else .Call(`_RcppSparse_columnSums`, A)
}

#' Keep a sparse matrix in GPU memory
#'
#' Uploads a \code{dgCMatrix} once; \code{columnSums()} on the returned handle runs on the
#' resident copy.  The handle is a copy: later changes of \code{A} are not seen.  The GPU
#' memory is released when the handle is garbage-collected, or at once by \code{gpuFree()}.
#'
#' @param A an object of class \code{dgCMatrix}
#' @param device GPU ordinal
#' @return external pointer of class \code{"gpuMatrix"}
#' @export
gpuMatrix <- function(A, device = as.integer(Sys.getenv("RCPPSPARSE_DEVICE", "0"))) {
    .Call(`_RcppSparse_gpuMatrix`, A, as.integer(device))
}

#' @rdname gpuMatrix
#' @param handle a \code{"gpuMatrix"}
#' @export
gpuFree <- function(handle) invisible(.Call(`_RcppSparse_gpuFree`, handle))
