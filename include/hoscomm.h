/*
 * hoscomm.h -- C ABI of libhoscomm.so: the collectives of the data-parallel step as plain-C entry points over RCCL
 * (SURVEY.md 8(b).6 "`hos_allreduce_*` wrappers over RCCL").
 *
 * The reference trains under PyTorch-Lightning DDP (3rd_Complete_HOSNeRF/run.py:173-190: `strategy = DDPStrategy`, one all-reduce
 * of the gradient buckets per step inside `loss.backward()`).  Here the flat gradient buffers are exchanged explicitly
 * (hosnerf_amd/train.py: allreduce_flat_grad); with torch.distributed the exchange has to stay OUTSIDE a captured hipGraph.  These
 * entry points enqueue the same RCCL collectives on a caller-given stream, so they can sit INSIDE the captured step (RCCL supports
 * stream capture) and a rank's whole step becomes one graph replay.  A separate library, so that libhosrender.so does not depend
 * on librccl.so.
 *
 * Conventions as in hosrender.h: device pointers + counts + a hipStream_t passed as void*; 0 on success, HOS_E_ARG (-1) on bad
 * arguments, 1000 + ncclResult_t on an RCCL error; never allocates device memory of its own, never synchronises.
 */
#ifndef HOSCOMM_H
#define HOSCOMM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HOS_COMM_ID_BYTES 128   /* = NCCL_UNIQUE_ID_BYTES */

typedef void* hos_comm_t;

/* Rank 0 creates the 128-byte rendezvous id and hands it to the other ranks out of band (hosnerf_amd/comm.py broadcasts it through
 * the torch.distributed store or a gloo group); every rank then joins with (id, nranks, rank) on its CURRENT hip device. */
int hos_comm_unique_id(void* id128);
int hos_comm_init(const void* id128, int nranks, int rank, hos_comm_t* comm);
int hos_comm_destroy(hos_comm_t comm);
int hos_comm_count(hos_comm_t comm, int* nranks);
int hos_comm_rank(hos_comm_t comm, int* rank);

/* In-place all-reduce of `count` floats at `buf`.  _sum: the reduction DDP performs on its buckets before dividing by the world
 * size; _avg: sum / nranks in the same pass (ncclAvg) -- what `allreduce_flat_grad` needs for the flat gradient buffers. */
int hos_allreduce_sum_f32(hos_comm_t comm, float* buf, int64_t count, void* stream);
int hos_allreduce_avg_f32(hos_comm_t comm, float* buf, int64_t count, void* stream);
/* In-place MAX of `count` unsigned 32-bit words: the fp16 range-guard word of the optimiser launch (hosrender.h: hos_adam_multi) --
 * a rank whose rays tripped the guard makes every rank skip the step, or the replicas would part for good. */
int hos_allreduce_max_u32(hos_comm_t comm, unsigned int* buf, int64_t count, void* stream);
/* recv [nranks * count_per_rank] <- every rank's send [count_per_rank] in rank order (inference: a frame's ray shards). */
int hos_allgather_f32(hos_comm_t comm, const float* send, float* recv, int64_t count_per_rank, void* stream);
/* Several spans in ONE RCCL group call (the human network's exchange: the volume gradient + the parameter spans outside the
 * volume decoder): bufs / counts are HOST arrays of n entries. */
int hos_allreduce_avg_f32_spans(hos_comm_t comm, float* const* bufs, const int64_t* counts, int n, void* stream);
/* Last words of a multi-rank job (bench.py's optional legs: a first execution of a collective path on N real devices can die in a
 * way no try/except sees -- abort() inside RCCL, a fault in a captured collective, the launcher's SIGTERM after another rank died).
 * hos_crash_line_set: keep a copy of `line` (len bytes, <= 1 MiB; len 0: say nothing) and install handlers for SIGSEGV, SIGBUS,
 * SIGABRT, SIGFPE, SIGILL and SIGTERM that write() it to file descriptor 1 and _exit(0) -- async-signal-safe calls only.
 * hos_crash_line_clear: restore the default dispositions.  The reference has no counterpart (Lightning's launcher just dies). */
int hos_crash_line_set(const char* line, int64_t len);
int hos_crash_line_clear(void);

#ifdef __cplusplus
}
#endif
#endif
