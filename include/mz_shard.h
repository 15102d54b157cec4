/* include/mz_shard.h -- a work list that exists on ONE rank of a process-per-GPU job, aligned on all of them (SURVEY.md section 8e:
 * "root-scatter / root-gather, variable sized ... ncclGroupStart(); for peer: ncclSend(...); ncclGroupEnd() on root and one ncclRecv per
 * peer; results gathered the same way in reverse").  The independent jobs this carries are the reference's tree drivers' --
 * /root/reference/tba.c:177-255 and auto_mz.c:101,113 start one multiz run per tree node, each a list of yama() calls
 * (mz_yama.h:22) that share nothing.
 *
 * What travels is what mz_yama_batch() puts on the PCIe link (include/mz_amd.h, link images): out, every rank's share of the jobs as
 * byte classes two per byte and band bounds as steps; back, a record per pair and the edit scripts at two bits per merged column; the
 * root assembles the merged columns from its own A and B.  HOW it travels is a table of five functions (mz_comm):
 *
 *   mz_comm_rccl_*      RCCL over xGMI (librccl is loaded when the first such comm is made; the library does not link it): grouped
 *                       ncclSend / ncclRecv of device buffers, one group per exchange -- every peer's link busy at once;
 *   mz_comm_loopback    every rank in this process, messages copied through a mailbox: the C code of the exchange runs without a
 *                       GPU and without a launcher (tests/test_shard_c.py);
 *   mz_comm_custom      the caller's own send / recv (multiz_amd/shard.py: torch.distributed "gloo" in the CPU tests).
 *
 * All functions return 0, or -1 with mz_last_error() set (mz_shard_gather: the number of pairs without a result).
 */
#ifndef MZ_SHARD_H
#define MZ_SHARD_H

#include "mz_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mz_comm mz_comm;
struct mz_comm {
    int rank, size;
    int device_buffers;                     /* 1: what send / recv move is device memory (alloc gives device memory) */
    void *self;
    int (*group_start)(mz_comm *);          /* sends and receives up to group_end() are issued together ... */
    int (*group_end)(mz_comm *);            /* ... and are complete when it returns */
    int (*send)(mz_comm *, const void *buf, size_t bytes, int peer);
    int (*recv)(mz_comm *, void *buf, size_t bytes, int peer);
    void *(*alloc)(mz_comm *, size_t bytes);                 /* a buffer send / recv can move (256-byte aligned) */
    void (*release)(mz_comm *, void *buf);
    int (*put)(mz_comm *, void *buf, const void *host, size_t bytes);      /* host -> such a buffer */
    int (*get)(mz_comm *, void *host, const void *buf, size_t bytes);      /* ... and back */
    void (*destroy)(mz_comm *);
};

/* RCCL: rank 0 makes an id (128 bytes: ncclUniqueId), everybody gets it by whatever means the job has, every rank joins.  The
 * communicator uses the GPU of the library's primary context and a stream of its own. */
int mz_comm_rccl_unique_id(void *id128);
int mz_comm_rccl_create(const void *id128, int rank, int size, mz_comm **comm);
/* `size` ranks inside this process (ranks[0 .. size)): a send is copied into a mailbox, a recv takes the oldest message of that peer
 * -- waiting for it up to ten seconds, so that the ranks may also be threads; a message of another size is an error */
int mz_comm_loopback(int size, mz_comm **ranks);
/* the caller's transport: host buffers, blocking send / recv (either may be NULL for group_start / group_end) */
int mz_comm_custom(int rank, int size, void *user,
                   int (*send)(void *user, const void *buf, size_t bytes, int peer), int (*recv)(void *user, void *buf, size_t bytes, int peer),
                   int (*group_start)(void *user), int (*group_end)(void *user), mz_comm **comm);
void mz_comm_free(mz_comm *comm);
/* `bytes` of a pattern from this rank to itself through the comm's send / recv in one group, compared on arrival (0: identical) */
int mz_comm_echo(mz_comm *comm, size_t bytes);

/* one rank's share of a scattered list */
typedef struct mz_shard mz_shard;
/* Root: n jobs, dealt over the ranks by cost (heaviest first, in a snake: every rank the same mix), every peer's share packed as a link
 * image and sent; the others: n = 0, jobs = NULL -- they receive.  Every rank gets its share (the root's own does not travel). */
int mz_shard_scatter(mz_comm *comm, int root, int n, const mz_job *jobs, mz_shard **shard);
/* this rank's share aligned where it lies (mz_link_plan + mz_link_finish: needs the GPU and device buffers); waits for the result */
int mz_shard_align(mz_shard *shard);
/* every rank's result image to the root, which assembles outs[0 .. n) of the scattered list, in the jobs' own order, from its own A
 * and B (release with mz_free_outs() -- also after -1: the shares assembled before the one that failed own their blocks, every other pair
 * is left MZ_E_DEVICE with cols == NULL).  Root: the jobs it scattered, outs; the others: NULL, NULL.  Returns the pairs without a result. */
int mz_shard_gather(mz_comm *comm, int root, mz_shard *shard, const mz_job *jobs, mz_out *outs);
void mz_shard_free(mz_comm *comm, mz_shard *shard);
/* what a share holds (for callers that align it themselves -- the CPU tests put the oracle there): its descriptor, pairs, the global
 * indices of its pairs, its image and exception block as host copies (library-owned, valid until the next call on this shard) */
const mz_link_desc *mz_shard_desc(const mz_shard *shard);
const int64_t *mz_shard_index(const mz_shard *shard);
int mz_shard_host_image(mz_comm *comm, mz_shard *shard, const void **image, const void **exc);
int mz_shard_set_result(mz_comm *comm, mz_shard *shard, const void *result, int64_t bytes);
/* band cells and pairs without a result of this rank's share (from its result image: after mz_shard_align / mz_shard_set_result) */
int mz_shard_totals(mz_comm *comm, const mz_shard *shard, int64_t *cells, int64_t *failed);
/* bytes this rank sent and received in its scatters / gathers so far (the root's are the exchange's totals) */
void mz_shard_traffic(int64_t *sent, int64_t *received);

/* ---------------------------------------------------------------- the same exchange in CHUNKS that overlap (round 6)
 *
 * scatter / align / gather above are three phases, each waiting for the one before on every rank.  mz_shard_run() is the whole
 * exchange as one call on every rank: the root deals the list over ranks x chunks (the same rule: every chunk of every rank the
 * same mix) and, step by step, packs chunk t+1 and assembles chunk t-3 on its host threads WHILE the transport moves chunk t down
 * and chunk t-2's result images up (one group each for the headers and the payloads: with RCCL every peer's link at once) WHILE
 * every rank's GPU aligns chunk t-1 of its share.  C chunks take C + 4 steps of max(host, link, GPU) instead of the three sums.
 *
 *   chunks     0: chosen from the list's size (about 4 Ki pairs a chunk and rank, at most 32; MZ_SHARD_CHUNKS overrides); the
 *              root's value counts, the others learn it from the first header
 *   align      NULL: the library's GPU path -- mz_link_plan + mz_link_finish where the chunk's image landed (a transport that moves
 *              host memory: the image goes up and the result comes down inside the step).  Else the caller's: called once per
 *              non-empty chunk of this rank's share with host copies of the image and its exception block; it hands the chunk's
 *              result image over with mz_shard_chunk_result(handle, ...) before it returns 0 (the CPU tests put the oracle there)
 *   times      (may be NULL) where this rank's time went; pairs / cells / failed: what this rank aligned
 * Root: n, jobs, outs as for mz_shard_gather (outs released with mz_free_outs(), also after -1); the others: 0, NULL, NULL.
 * Returns the pairs without a result on the root, 0 on the others, -1 on an error of this rank (mz_last_error()).  A chunk that
 * fails on one rank's GPU travels as failed -- its pairs stay MZ_E_DEVICE on the root -- and the exchange runs to its end. */
typedef struct mz_shard_times {
    double pack_s, comm_s, align_s, assemble_s, wall_s;     /* sums over the steps (they overlap: wall_s is less than their sum) */
    int chunks, steps;
    int64_t pairs, cells, failed;
} mz_shard_times;
typedef int (*mz_shard_align_fn)(void *user, int chunk, const mz_link_desc *desc, const void *image, const void *exc, void *handle);
int mz_shard_chunk_result(void *handle, const void *result, int64_t bytes);
int mz_shard_run(mz_comm *comm, int root, int n, const mz_job *jobs, mz_out *outs, int chunks, mz_shard_align_fn align, void *user,
                 mz_shard_times *times);

#ifdef __cplusplus
}
#endif
#endif
