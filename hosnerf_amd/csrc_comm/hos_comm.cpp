// libhoscomm.so: include/hoscomm.h over RCCL.  Host code only (no kernels): compiled with hipcc for the hip runtime headers.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <string.h>

#include "hoscomm.h"

static inline int rc_of(ncclResult_t r) { return r == ncclSuccess ? 0 : 1000 + (int)r; }

extern "C" int hos_comm_unique_id(void* id128) {
    if (!id128) return -1;
    ncclUniqueId id;
    const ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) return rc_of(r);
    static_assert(sizeof(id) == HOS_COMM_ID_BYTES, "id size");
    memcpy(id128, &id, sizeof(id));
    return 0;
}

extern "C" int hos_comm_init(const void* id128, int nranks, int rank, hos_comm_t* comm) {
    if (!id128 || !comm || nranks <= 0 || rank < 0 || rank >= nranks) return -1;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t c = nullptr;
    const ncclResult_t r = ncclCommInitRank(&c, nranks, id, rank);
    if (r != ncclSuccess) return rc_of(r);
    *comm = c;
    return 0;
}

extern "C" int hos_comm_destroy(hos_comm_t comm) {
    if (!comm) return -1;
    return rc_of(ncclCommDestroy(static_cast<ncclComm_t>(comm)));
}

extern "C" int hos_comm_count(hos_comm_t comm, int* nranks) {
    if (!comm || !nranks) return -1;
    return rc_of(ncclCommCount(static_cast<ncclComm_t>(comm), nranks));
}

extern "C" int hos_comm_rank(hos_comm_t comm, int* rank) {
    if (!comm || !rank) return -1;
    return rc_of(ncclCommUserRank(static_cast<ncclComm_t>(comm), rank));
}

static int allreduce(hos_comm_t comm, float* buf, int64_t count, ncclRedOp_t op, void* stream) {
    if (!comm || !buf || count <= 0) return -1;
    return rc_of(ncclAllReduce(buf, buf, (size_t)count, ncclFloat32, op, static_cast<ncclComm_t>(comm), static_cast<hipStream_t>(stream)));
}

extern "C" int hos_allreduce_sum_f32(hos_comm_t comm, float* buf, int64_t count, void* stream) { return allreduce(comm, buf, count, ncclSum, stream); }
extern "C" int hos_allreduce_avg_f32(hos_comm_t comm, float* buf, int64_t count, void* stream) { return allreduce(comm, buf, count, ncclAvg, stream); }

// in-place MAX over the ranks of `count` unsigned 32-bit words (the fp16 range-guard word: every rank must take the same decision)
extern "C" int hos_allreduce_max_u32(hos_comm_t comm, unsigned int* buf, int64_t count, void* stream) {
    if (!comm || !buf || count <= 0) return -1;
    return rc_of(ncclAllReduce(buf, buf, (size_t)count, ncclUint32, ncclMax, static_cast<ncclComm_t>(comm), static_cast<hipStream_t>(stream)));
}

extern "C" int hos_allgather_f32(hos_comm_t comm, const float* send, float* recv, int64_t count_per_rank, void* stream) {
    if (!comm || !send || !recv || count_per_rank <= 0) return -1;
    return rc_of(ncclAllGather(send, recv, (size_t)count_per_rank, ncclFloat32, static_cast<ncclComm_t>(comm), static_cast<hipStream_t>(stream)));
}

extern "C" int hos_allreduce_avg_f32_spans(hos_comm_t comm, float* const* bufs, const int64_t* counts, int n, void* stream) {
    if (!comm || !bufs || !counts || n <= 0) return -1;
    for (int i = 0; i < n; ++i) if (!bufs[i] || counts[i] <= 0) return -1;
    ncclResult_t r = ncclGroupStart();
    if (r != ncclSuccess) return rc_of(r);
    for (int i = 0; i < n; ++i) {
        r = ncclAllReduce(bufs[i], bufs[i], (size_t)counts[i], ncclFloat32, ncclAvg, static_cast<ncclComm_t>(comm), static_cast<hipStream_t>(stream));
        if (r != ncclSuccess) { ncclGroupEnd(); return rc_of(r); }
    }
    return rc_of(ncclGroupEnd());
}
