// libhoscomm.so: include/hoscomm.h over RCCL.  Host code only (no kernels): compiled with hipcc for the hip runtime headers.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

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


// ---- last words (include/hoscomm.h)
static char g_crash_line[1 << 20];
static volatile long g_crash_len = 0;
static const int g_crash_signals[] = {SIGSEGV, SIGBUS, SIGABRT, SIGFPE, SIGILL, SIGTERM};

static void hos_crash_handler(int) {
    long done = 0;
    const long n = g_crash_len;
    while (done < n) {
        const ssize_t w = write(1, g_crash_line + done, (size_t)(n - done));
        if (w <= 0) break;
        done += w;
    }
    _exit(0);
}

extern "C" int hos_crash_line_set(const char* line, int64_t len) {
    if (len < 0 || len > (int64_t)sizeof(g_crash_line) || (len > 0 && !line)) return -1;
    g_crash_len = 0;
    if (len > 0) memcpy(g_crash_line, line, (size_t)len);
    g_crash_len = (long)len;
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_handler = hos_crash_handler;
    sigemptyset(&sa.sa_mask);
    for (int s : g_crash_signals) sigaction(s, &sa, nullptr);
    return 0;
}

extern "C" int hos_crash_line_clear(void) {
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_handler = SIG_DFL;
    sigemptyset(&sa.sa_mask);
    for (int s : g_crash_signals) sigaction(s, &sa, nullptr);
    g_crash_len = 0;
    return 0;
}
