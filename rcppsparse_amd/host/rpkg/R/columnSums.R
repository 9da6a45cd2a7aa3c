#' Column sums of a sparse matrix on the GPU
#'
#' @param A an object of class \code{dgCMatrix}
#' @return numeric vector of length \code{ncol(A)}
#' @export
columnSums <- function(A) .Call(`_RcppSparse_columnSums`, A)
