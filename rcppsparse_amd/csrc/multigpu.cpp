// multigpu.cpp -- column-range partitioner (pure integer, host) and the RCCL
// gatherv of per-shard sums.  Nothing like this exists in the reference (it has
// no distributed code at all); columns are independent
// units of reference src/example.cpp:28, so contiguous column ranges shard the
// path with exactly one exchange step: the gather of disjoint output slices.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>

#include "../../include/rcppsparse_hip.h"

static_assert(RSP_UNIQUE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");

namespace rsp {
int fail(int code, const char* fmt, ...);   // capi.hip: sets rsp_last_error() text
}
using rsp::fail;

struct rsp_comm {
    ncclComm_t comm;
    int nranks, rank, device;
};

extern "C" {

int rsp_partition_columns(const int32_t* p, int32_t ncol, int32_t nparts, int32_t* bounds) {
    if (!p || !bounds || ncol < 0 || nparts <= 0)
        return fail(RSP_ERR_BAD_ARG, "bad argument to rsp_partition_columns");
    const int64_t nnz = p[ncol];
    bounds[0] = 0;
    for (int32_t k = 1; k < nparts; ++k) {
        // first column c with p[c] >= k*nnz/nparts  (lower_bound on p[0..ncol])
        const int64_t target = (int64_t)(((__int128)k * nnz) / nparts);
        int32_t lo = 0, hi = ncol;   // answer in [lo, hi]; p[ncol] = nnz >= target
        while (lo < hi) {
            const int32_t mid = lo + (hi - lo) / 2;
            if (p[mid] >= target) hi = mid; else lo = mid + 1;
        }
        bounds[k] = lo < bounds[k - 1] ? bounds[k - 1] : lo;
    }
    bounds[nparts] = ncol;
    return RSP_OK;
}

int rsp_rebase_offsets(const int32_t* p, int32_t c0, int32_t c1, int32_t* p_local) {
    if (!p || !p_local || c0 < 0 || c1 < c0)
        return fail(RSP_ERR_BAD_ARG, "bad argument to rsp_rebase_offsets");
    const int32_t base = p[c0];
    for (int32_t j = 0; j <= c1 - c0; ++j) p_local[j] = p[c0 + j] - base;
    return RSP_OK;
}

int rsp_comm_unique_id(void* id_bytes) {
    if (!id_bytes) return fail(RSP_ERR_BAD_ARG, "id_bytes is null");
    ncclUniqueId id;
    ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) return fail(RSP_ERR_RCCL, "ncclGetUniqueId: %s", ncclGetErrorString(r));
    memcpy(id_bytes, &id, sizeof(id));
    return RSP_OK;
}

int rsp_comm_init(const void* id_bytes, int nranks, int rank, int device, rsp_comm_t* comm) {
    if (!id_bytes || !comm || nranks <= 0 || rank < 0 || rank >= nranks)
        return fail(RSP_ERR_BAD_ARG, "bad argument to rsp_comm_init");
    *comm = nullptr;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return fail(RSP_ERR_HIP, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof(id));
    rsp_comm* c = new (std::nothrow) rsp_comm();
    if (!c) return fail(RSP_ERR_ALLOC, "out of host memory");
    ncclResult_t r = ncclCommInitRank(&c->comm, nranks, id, rank);
    if (r != ncclSuccess) {
        delete c;
        return fail(RSP_ERR_RCCL, "ncclCommInitRank: %s", ncclGetErrorString(r));
    }
    c->nranks = nranks;
    c->rank = rank;
    c->device = device;
    *comm = c;
    return RSP_OK;
}

int rsp_comm_gatherv(rsp_comm_t c, const double* d_send, int64_t send_count, double* d_recv,
                     const int64_t* counts, const int64_t* displs, int root, void* stream) {
    if (!c || root < 0 || root >= c->nranks || send_count < 0)
        return fail(RSP_ERR_BAD_ARG, "bad argument to rsp_comm_gatherv");
    hipStream_t s = (hipStream_t)stream;
    ncclResult_t r = ncclGroupStart();
    if (r != ncclSuccess) return fail(RSP_ERR_RCCL, "ncclGroupStart: %s", ncclGetErrorString(r));
    if (c->rank == root) {
        if (!counts || !displs || !d_recv) {
            (void)ncclGroupEnd();
            return fail(RSP_ERR_BAD_ARG, "root needs d_recv, counts and displs");
        }
        for (int k = 0; k < c->nranks && r == ncclSuccess; ++k) {
            if (k == root) {
                // own slice: device-to-device copy on the same stream (skipped when in place)
                if (counts[k] > 0 && d_send != d_recv + displs[k]) {
                    hipError_t e = hipMemcpyAsync(d_recv + displs[k], d_send, (size_t)counts[k] * 8,
                                                  hipMemcpyDeviceToDevice, s);
                    if (e != hipSuccess) {
                        (void)ncclGroupEnd();
                        return fail(RSP_ERR_HIP, "root copy: %s", hipGetErrorString(e));
                    }
                }
            } else if (counts[k] > 0) {
                r = ncclRecv(d_recv + displs[k], (size_t)counts[k], ncclDouble, k, c->comm, s);
            }
        }
    } else if (send_count > 0) {
        r = ncclSend(d_send, (size_t)send_count, ncclDouble, root, c->comm, s);
    }
    ncclResult_t r2 = ncclGroupEnd();
    if (r != ncclSuccess) return fail(RSP_ERR_RCCL, "ncclSend/Recv: %s", ncclGetErrorString(r));
    if (r2 != ncclSuccess) return fail(RSP_ERR_RCCL, "ncclGroupEnd: %s", ncclGetErrorString(r2));
    return RSP_OK;
}

int rsp_comm_destroy(rsp_comm_t c) {
    if (!c) return RSP_OK;
    (void)hipSetDevice(c->device);
    ncclResult_t r = ncclCommDestroy(c->comm);
    delete c;
    if (r != ncclSuccess) return fail(RSP_ERR_RCCL, "ncclCommDestroy: %s", ncclGetErrorString(r));
    return RSP_OK;
}

}  // extern "C"
