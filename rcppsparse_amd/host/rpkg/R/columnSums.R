#' Column sums of a sparse matrix on the GPU
#'
#' @param A an object of class \code{dgCMatrix}, or a handle made by \code{\link{gpuMatrix}} (the matrix is
#'   then already in GPU memory and is not transferred)
#' @return numeric vector of length \code{ncol(A)}
#' @export
columnSums <- function(A) {
    if (inherits(A, "gpuMatrixMulti")) .Call(`_RcppSparse_gpuMultiReduce`, A, 0L)
    else if (inherits(A, "gpuMatrix")) .Call(`_RcppSparse_gpuColumnSums`, A)
    else .Call(`_RcppSparse_columnSums`, A)
}

#' Which path answers columnSums()
#'
#' \code{columnSums()} on a \code{dgCMatrix} runs on the GPU whenever one is visible.  On a machine without
#' any GPU it answers with the same column loop on the host, like the CPU original -- unless
#' \code{options(RcppSparse.require_gpu = TRUE)} (or \code{RCPPSPARSE_REQUIRE_GPU=1} in the environment) is
#' set, in which case it is an error.
#' @param last \code{FALSE}: the path a call would take now; \code{TRUE}: the path that answered the most
#'   recent call
#' @return \code{"hip"}, \code{"cpu"} or \code{"none"}
#' @export
columnSumsBackend <- function(last = FALSE) .Call(`_RcppSparse_columnSumsBackend`, as.integer(isTRUE(last)))

#' Keep a sparse matrix in GPU memory
#'
#' Uploads a \code{dgCMatrix} once; \code{columnSums()}, \code{gpuColMeans()}, \code{gpuRowSums()},
#' \code{gpuRowMeans()} and \code{gpuCrossprod()} on the returned handle run on the resident copy.  The handle
#' is a copy: later changes of \code{A} are not seen.  The GPU memory is released when the handle is
#' garbage-collected, or at once by \code{gpuFree()}.  With several \code{devices} the columns are cut into
#' nnz-balanced ranges, one resident shard per entry (class \code{"gpuMatrixMulti"}; no \code{gpuCrossprod}).
#'
#' @param A an object of class \code{dgCMatrix}
#' @param device GPU ordinal
#' @param devices GPU ordinals for a matrix spread over several GPUs of the node (\code{NULL}: one GPU)
#' @return external pointer of class \code{"gpuMatrix"} or \code{"gpuMatrixMulti"}
#' @export
gpuMatrix <- function(A, device = as.integer(Sys.getenv("RCPPSPARSE_DEVICE", "0")), devices = NULL) {
    if (length(devices) > 1L) .Call(`_RcppSparse_gpuMatrixMulti`, A, as.integer(devices))
    else .Call(`_RcppSparse_gpuMatrix`, A, as.integer(if (length(devices) == 1L) devices else device))
}

.gpuReduce <- function(handle, what) {
    if (inherits(handle, "gpuMatrixMulti")) .Call(`_RcppSparse_gpuMultiReduce`, handle, what)
    else if (inherits(handle, "gpuMatrix")) .Call(`_RcppSparse_gpuReduce`, handle, what)
    else stop("not a gpuMatrix handle")
}

#' Column means, row sums, row means and crossprod of a GPU-resident matrix
#'
#' The device forms of the C++ class's \code{colMeans()}, \code{rowSums()}, \code{rowMeans()} and
#' \code{crossprod()} on a handle made by \code{\link{gpuMatrix}}.
#' @param handle a \code{"gpuMatrix"} (or \code{"gpuMatrixMulti"}, except for \code{gpuCrossprod})
#' @export
gpuColMeans <- function(handle) .gpuReduce(handle, 1L)
#' @rdname gpuColMeans
#' @export
gpuRowSums <- function(handle) .gpuReduce(handle, 2L)
#' @rdname gpuColMeans
#' @export
gpuRowMeans <- function(handle) .gpuReduce(handle, 3L)
#' @rdname gpuColMeans
#' @export
gpuCrossprod <- function(handle) {
    if (!inherits(handle, "gpuMatrix")) stop("gpuCrossprod needs a single-GPU gpuMatrix handle")
    .Call(`_RcppSparse_gpuCrossprod`, handle)
}

#' @export
dim.gpuMatrix <- function(x) attr(x, "Dim")
#' @export
dim.gpuMatrixMulti <- function(x) attr(x, "Dim")

#' @rdname gpuMatrix
#' @param handle a handle made by \code{gpuMatrix}
#' @export
gpuFree <- function(handle) {
    if (inherits(handle, "gpuMatrixMulti")) invisible(.Call(`_RcppSparse_gpuFreeMulti`, handle))
    else invisible(.Call(`_RcppSparse_gpuFree`, handle))
}
