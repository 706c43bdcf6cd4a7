/*
 * sip_lqr_amd_rccl.h -- multi-GPU side of the batched LQR path: the exchange of
 * the feedback gains (K, k) between the GPUs of one node over RCCL / xGMI.
 *
 * Problem instances are independent (no shared state between the reference's
 * LQR / Workspace pairs, lqr.hpp:189-199), so the batch is block-partitioned
 * over the GPUs with no data-path collective; the one exchange step is the
 * all-gather of every shard's gains.  The reference
 * has no multi-GPU code; this header completes the C ABI of sip_lqr_amd.h for
 * callers that are not torch programs.  It lives in its own library
 * (libsip_lqr_amd_rccl.so, links librccl) so that libsip_lqr_amd.so itself
 * carries no communication dependency.
 *
 * Two process models:
 *   - one process per GPU (what bench.py does through torch.distributed): the
 *     caller owns its ncclComm_t (ncclCommInitRank) and calls
 *     sip_lqr_all_gather_gains();
 *   - one process driving several GPUs: sip_lqr_group_create() builds the
 *     communicators (ncclCommInitAll) and sip_lqr_group_all_gather_gains()
 *     issues the collective for all of them inside one ncclGroupStart/End.
 */
#ifndef SIP_LQR_AMD_RCCL_H
#define SIP_LQR_AMD_RCCL_H

#include "sip_lqr_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sip_lqr_group sip_lqr_group;

/* devices[i]: HIP device ordinal of rank i; ndev >= 1. */
int sip_lqr_group_create(int ndev, const int *devices, sip_lqr_group **group);
void sip_lqr_group_destroy(sip_lqr_group *group);
int sip_lqr_group_size(const sip_lqr_group *group);

/* Rank i contributes d_gains[i] (sip_lqr_gains_bytes(plans[i]) bytes, the same on every rank) and
 * receives the shards of all ranks, rank-major, in d_all_gains[i] (ndev times that); the copy for
 * rank i is enqueued on streams[i] (hipStream_t), behind the sweep that produced the gains. */
int sip_lqr_group_all_gather_gains(sip_lqr_group *group, const sip_lqr_plan *const *plans,
                                   const void *const *d_gains, void *const *d_all_gains,
                                   void *const *streams);

/* One process per GPU: `nccl_comm` is the caller's ncclComm_t. */
int sip_lqr_all_gather_gains(const sip_lqr_plan *plan, void *nccl_comm, const void *d_gains,
                             void *d_all_gains, void *stream);

/* ---- chunk-pipelined exchange -----------------------------------------
 * The shard of a rank is cut into `num_chunks` contiguous problem ranges (sip_lqr_gains_chunk_range:
 * equal sizes, the last ones one problem smaller when it does not divide; identical on every rank),
 * and chunk c of every rank is gathered by its own collective, which the caller enqueues behind
 * whatever produced that chunk's gains (an event behind the sweep, or behind one of several
 * sub-batch launches) while later chunks / the next sweep still compute.  Collectives must be
 * issued in the same order on every rank: enqueue the chunks of a sweep in chunk order.
 * Gathered layout: chunk-major, the bytes of (rank r, chunk c) at sip_lqr_gains_chunk_offset(): what
 * one ncclAllGather per chunk produces with no re-packing. */
int sip_lqr_gains_chunk_range(const sip_lqr_plan *plan, int chunk, int num_chunks, int64_t *first_problem,
                              int64_t *num_problems);
size_t sip_lqr_gains_chunk_offset(const sip_lqr_plan *plan, int nranks, int rank, int chunk, int num_chunks);
/* `nranks` must be the size of `nccl_comm` (checked: the chunk's place in the gathered buffer follows from it). */
int sip_lqr_all_gather_gains_chunk(const sip_lqr_plan *plan, void *nccl_comm, int nranks, const void *d_gains,
                                   void *d_all_gains, int chunk, int num_chunks, void *stream);
int sip_lqr_group_all_gather_gains_chunk(sip_lqr_group *group, const sip_lqr_plan *const *plans,
                                         const void *const *d_gains, void *const *d_all_gains, int chunk,
                                         int num_chunks, void *const *streams);

/* One-process groups: the same exchange as direct peer copies over the xGMI mesh, one copy per
 * (rank, peer) pair on a stream of its own, so that the seven links of a GPU carry its shard to the
 * seven peers at the same time (a ring would push the shard through one link seven times).  Rank-major
 * layout, as sip_lqr_group_all_gather_gains.  Each copy waits for the work enqueued on streams[i] when
 * the call is made; streams[i] then waits for the copies INTO rank i and for the copies OUT of rank i,
 * so both buffers follow ncclAllGather's stream semantics: behind the call on streams[i], d_all_gains[i]
 * is complete and d_gains[i] may be rewritten.
 * Needs peer access between all devices of the group (enabled by sip_lqr_group_create where the
 * hardware offers it; SIP_LQR_ERR_UNSUPPORTED otherwise). */
int sip_lqr_group_all_gather_gains_p2p(sip_lqr_group *group, const sip_lqr_plan *const *plans,
                                       const void *const *d_gains, void *const *d_all_gains,
                                       void *const *streams);

/* Chunk form with a completion stream of the caller's choice: chunk `chunk` of `num_chunks` of every
 * rank, chunk-major layout (sip_lqr_gains_chunk_offset).  streams[i] is the stream whose work produces
 * rank i's chunk (the copies wait for what it holds at call time); done_streams[i] (NULL array: the
 * same as streams) is the stream that is made to wait for the copies into AND out of rank i, and whose
 * work at call time the copies into rank i wait for (consumers of the previous contents of the gathered
 * buffer live there).  With a done stream of its own the compute stream is not held up by the exchange:
 * the next sweep overlaps it, provided the caller double-buffers d_gains (or makes the compute stream
 * wait for the done stream before it rewrites them).  Chunks of one pair travel in call order. */
int sip_lqr_group_all_gather_gains_p2p_chunk(sip_lqr_group *group, const sip_lqr_plan *const *plans,
                                             const void *const *d_gains, void *const *d_all_gains, int chunk,
                                             int num_chunks, void *const *streams, void *const *done_streams);

#ifdef __cplusplus
}
#endif
#endif
