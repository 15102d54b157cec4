/* mz_ctx.h -- internal to libmzamd: the per-GPU context and the helpers shared by mz_host.c (life cycle, scores,
 * device-resident API, pre_yama batches) and mz_batch.c (the host-buffer batch pipeline).  Nothing here is exported. */
#ifndef MZAMD_MZ_CTX_H
#define MZAMD_MZ_CTX_H

#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <pthread.h>
#include <stddef.h>
#include <stdint.h>
#include <time.h>

#include "mz_device.h"

#define MZ_INTERNAL __attribute__((visibility("hidden")))
#define MZ_INTERNAL_DECL __attribute__((visibility("hidden")))

typedef struct gbuf { void *p; size_t cap; } gbuf;

#define MZ_SLICES 4                        /* (number of helper events) */
#define MZ_SETS 10                         /* buffer sets of the chunk pipeline (mz_batch.c): being packed, uploading + planning,
                                            * computing, copying back, being unpacked -- and one of slack */
/* mz_preyama_batch()'s own buffers of a set.  Device: staging image (header + class nibbles), text expanded to a byte per class, the
 * pools A / B of the first stage, its band pools, k_pre's scratch, per-merge arrays of both stages, merged columns of the first and of
 * the second stage, the second stage's A pool (top rows), band pools, plan arrays and workspaces, results.  Pinned host: staging,
 * results, the second plan's totals. */
enum { MZ_PD_IN, MZ_PD_TXT, MZ_PD_COLS, MZ_PD_BAND, MZ_PD_SCR, MZ_PD_META, MZ_PD_OUT1, MZ_PD_OUT2, MZ_PD_A2, MZ_PD_BAND2, MZ_PD_PLAN2, MZ_PD_TB2,
       MZ_PD_SCRIPT2, MZ_PD_PREP2, MZ_PD_RES, MZ_PD_N };
enum { MZ_PH_IN, MZ_PH_RES, MZ_PH_TOT2, MZ_PH_N };
#define MZ_QS 4                            /* streams of one kind in the chunk pipelines at most (two; four for calls of few long pairs: mzi_flow_wide) */
#define MZ_QALL 24                         /* ... and in all, fillers included (mz_flow.c creates them in rounds of four) */
#define MZ_FLOW_STAGES 4                   /* stage threads of a chunk pipeline at most */
#define MZ_WS_MAX 8                        /* workspaces remembered by mz_dev_run_async() */
#define MZ_MAX_DEV 16
#define MZ_MULTI_MIN 2048                  /* pairs per GPU below which dealing a batch out is not worth a thread */
#define MZ_COPY_THREADS 24                 /* host threads of the pack / unpack loops: enough to saturate memory bandwidth; waking a
                                            * whole 256-thread pool for a 2 ms loop costs more than it saves (and was seen to stall
                                            * for 70-100 ms now and then) */

struct mz_pipe;

/* a persistent helper thread of the chunk pipeline (its OpenMP team lives as long as it does) */
typedef struct mz_worker {
    pthread_t th;
    pthread_mutex_t mu;
    pthread_cond_t cv;
    int started, quit, busy;
    void (*fn)(void *);
    void *job;
} mz_worker;

/* mz_pool.c: start the helper threads w[0..n) (idempotent; -1 when a thread cannot be had), hand one a job -- fn(job) runs on it; the
 * worker is free again when fn returns, and mzi_worker_give() waits for that --, end them */
MZ_INTERNAL_DECL int mzi_workers_start(mz_worker *w, int n);
MZ_INTERNAL_DECL void mzi_worker_give(mz_worker *w, void (*fn)(void *), void *job);
MZ_INTERNAL_DECL void mzi_workers_end(mz_worker *w, int n);

/* Everything the library holds on ONE GPU.  g_dev[0] is the primary context: the device-resident API (mz_dev_*)
 * and single-GPU runs live there.  mz_init_multi() / MZ_NGPU add contexts on further GPUs, each driven by its own
 * host thread when mz_yama_batch() deals a large batch out over them (block pairs are
 * independent, so the shards never talk to each other). */
typedef struct mz_ctx {
    int ready;
    int device;
    hipStream_t stream;
    hipStream_t stream2;                   /* pipelined form: traceback walk + emit of batch k beside the DP of batch k+1 */
    hipStream_t stream3;                   /* pipelined form: plan of batch k+1 beside the DP of batch k */
    hipStream_t stream_dp[4];              /* pipelined form, small batches: the DPs of consecutive batches side by side */
    unsigned dp_turn;
    hipEvent_t ev[5];
    hipEvent_t evs[MZ_SLICES + 1];
    int scores_ok;                         /* the device's copy of the score model is current */
    /* grow-only buffers of the host-buffer path, one of each per set:
     *   h_in / d_in    staging block (header, band steps, class nibbles of A and B) and its device image
     *   h_exc / d_exc  band bounds that do not fit a nibble per step
     *   d_cols, d_band the pools the kernels read (expanded on the device)
     *   d_plan, h_tot  plan arrays, the plan's totals
     *   d_tb, d_script, d_prep   workspaces
     *   d_res / h_res  results: header, a record per pair, the packed edit scripts */
    gbuf h_in[MZ_SETS], d_in[MZ_SETS], h_exc[MZ_SETS], d_exc[MZ_SETS], d_cols[MZ_SETS], d_band[MZ_SETS], d_plan[MZ_SETS], h_tot[MZ_SETS],
         d_tb[MZ_SETS], d_script[MZ_SETS], d_prep[MZ_SETS], d_res[MZ_SETS], h_res[MZ_SETS];
    /* mz_preyama_batch() (mz_prebatch.c), per set, beside d_plan / d_tb / d_script / d_prep / h_tot / the stream and events above (its first
     * stage uses those): device buffers pd[set][PD_*], pinned host buffers ph[set][PH_*]; `pplan2`: the second stage's plan is through */
    gbuf pd[MZ_SETS][MZ_PD_N], ph[MZ_SETS][MZ_PH_N];
    hipEvent_t pplan2[MZ_SETS];
    hipEvent_t ptime[MZ_SETS][8];          /* MZ_TIMING=2: before the upload, uploaded, k_pre done, planned, first DP done, first emit done, (second stage done,) k_fin done */
    int ptime_ready;
    hipStream_t bstream[MZ_SETS];          /* bstream[0] = `stream`: a call of one chunk (the drop-in yama()) runs there */
    /* The chunk pipelines' streams (mz_flow.c; created on first use).  Chunk k of a call: its staging block -> device copy (a kernel:
     * mzk_link_copy) on qc; expansion and plan on qf[0]; its DP kernels on qd[k % 2] -- back to back with the DPs of the chunks
     * before and after it, as the device-resident pipeline runs them (mz_dev_run_async), forking onto the slot's own lanes when the
     * chunk has several kinds of pairs; walk, script packing and the results -> host copy on qt[k % nt], behind the DP's event (bdp).
     * WHICH hardware queue a stream gets matters (mz_flow.c: the queues of one pipe of the command processor hold each other up). */
    hipStream_t qf[MZ_QS], qd[MZ_QS], qt[MZ_QS], qc, qall[MZ_QALL];
    mz_dp_lanes qlane[MZ_QS];
    int nq, nf, nt, nqall, lanes_made, nq_wide;     /* DP slots (chunk k: qd[k % nq]), front streams (qf[k % nf]), tail streams (qt[k % nt]); streams created; the lanes exist */
    hipEvent_t bdone[MZ_SETS], bplan[MZ_SETS], bprep[MZ_SETS], bdp[MZ_SETS], bcopy[MZ_SETS];    /* the chunk's last kernel; its plan's totals; its prep records; its DP kernels; its staging block on the device */
    hipEvent_t btime[MZ_SETS][6];          /* MZ_TIMING=2: start, uploaded, planned, DP done, results packed, copied back */
    int btime_ready;
    struct { const void *key; hipEvent_t done; int used; } ws[MZ_WS_MAX];
    int ws_victim;
    int copy_threads;                      /* host threads of this context's pack / unpack loops */
    mz_worker fworker[MZ_FLOW_STAGES];     /* the chunk pipeline's stage threads (mz_flow.c), shared by the two host paths (calls are serialised) */
} mz_ctx;

extern MZ_INTERNAL mz_ctx g_dev[MZ_MAX_DEV];
extern MZ_INTERNAL int g_ndev;
extern MZ_INTERNAL pthread_mutex_t g_big;
extern MZ_INTERNAL int g_hint_gen;
#define G (g_dev[0])

MZ_INTERNAL int mzi_set_err(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
#define HIPCK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) \
    return mzi_set_err("%s failed: %s", #call, hipGetErrorString(e_)); } while (0)

MZ_INTERNAL int mzi_dev_reserve(gbuf *b, size_t need);
MZ_INTERNAL int mzi_host_reserve(gbuf *b, size_t need);
MZ_INTERNAL int mzi_lazy_stream(hipStream_t *s);
MZ_INTERNAL unsigned mzi_event_flags(void);              /* of the events nobody takes times from: no timing, release to the DEVICE (mz_host.c) */
MZ_INTERNAL int mzi_flow_streams(mz_ctx *X);              /* mz_flow.c: the chunk streams and their lanes, on first use */
MZ_INTERNAL int mzi_flow_wide(mz_ctx *X);                 /* ... DP and tail streams 2 and 3, for calls of few long pairs (four chunks' DPs side by side) */
MZ_INTERNAL int mzi_flow_lanes(mz_ctx *X);                /* ... the DP streams' lanes, when a chunk with several kinds of pairs needs them */
MZ_INTERNAL void mzi_flow_sync(mz_ctx *X);                /* ... all of them synchronised (after an error) */
MZ_INTERNAL int mzi_ensure_init(void);
extern MZ_INTERNAL __thread int mzi_warm_thread;     /* set in the thread of mz_warm_start(): its batch prints no MZ_TIMING line */
MZ_INTERNAL int mzi_sync_scores(void);
MZ_INTERNAL int mzi_timing(void);                       /* MZ_TIMING, parsed once: 0 quiet, 1 per call, 2 per chunk */
MZ_INTERNAL void mzi_workers_stop(mz_ctx *X);           /* mz_batch.c: ctx_close() ends the context's helper threads */
MZ_INTERNAL int mzi_deal_snake(int n, const double *weight, int use, int *owner, int *where, int *cnt, int *start);    /* mz_batch.c */
MZ_INTERNAL int64_t mzi_result_image_min(int n);         /* mz_batch.c: bytes of the header and the records of a result image of n pairs */
MZ_INTERNAL void mzi_link_forget(void);                 /* mz_batch.c: a planned link image (set 0 of the primary context) is gone */
MZ_INTERNAL int mzi_pre_on_ctx(mz_ctx *X, int n, const mz_prejob *jobs, mz_preout *outs, int64_t stats[3]);   /* mz_prebatch.c */

/* mz_pool.c: the host threads of the batch pipeline (no OpenMP there: see the file), recycled result blocks */
typedef void (*mz_pfn)(void *ctx, int lo, int hi);
MZ_INTERNAL void mzi_parallel_for(int n, int grain, mz_pfn fn, void *ctx);
/* a loop that is posted and not waited for: fn(ctx, lo, hi) over [0, n) in pieces of `grain` on the pool's threads, then done(arg)
 * once, on the thread that ran the last piece (next / pending / link are the pool's) */
#define MZ_HEDGE_PIECES 512
typedef struct mz_ajob {
    mz_pfn fn; void *ctx; int n, grain;
    void (*done)(void *arg); void *arg;
    int hedge;                             /* 1: a piece may be run twice (mz_pool.c: a late one is handed out again) -- pieces must be idempotent */
    int next, pending, active, hedged, npiece; struct mz_ajob *link, *olink;
    unsigned char state[MZ_HEDGE_PIECES]; double t_start[MZ_HEDGE_PIECES];
    double t_sum; int n_done;              /* first runs that came back: their durations, their number (what "late" is measured against) */
    double skip_until;                     /* no piece of this loop can be late before then: the search for late pieces passes it by (mz_pool.c) */
} mz_ajob;
MZ_INTERNAL void mzi_post(mz_ajob *job);
MZ_INTERNAL int mzi_job_quiet(mz_ajob *job);                            /* wait until no thread is in a piece of a (complete) posted loop; pieces run twice */
MZ_INTERNAL void mzi_help_until(int (*ready)(void *), void *arg);    /* the caller works on posted pieces until ready(arg) (evaluated under the pool's lock) */
MZ_INTERNAL void mzi_pool_kick(void);                                  /* ... whoever makes ready() true calls this afterwards */
MZ_INTERNAL int mzi_pool_threads(void);
MZ_INTERNAL int mzi_cpu_budget(void);
MZ_INTERNAL void mzi_pool_stop(void);
MZ_INTERNAL void *mzi_block_get(size_t need);
MZ_INTERNAL void mzi_block_put(void *p);
MZ_INTERNAL void mzi_blocks_drop(void);

/* mz_flow.c: the chunk pipeline of mz_yama_batch() and mz_preyama_batch().  A call is cut into chunks; chunk k lives in buffer set
 * k % MZ_SETS and goes through
 *   cut      (the calling thread)   the client lays the chunk out in its set and describes its packing as a loop, which is POSTED to the
 *                                   pool (mz_pool.c) -- the caller does not wait for it: it cuts the next chunk, and works in the pool
 *                                   itself when it has nothing to cut;
 *   stage 1 .. nstage (a persistent thread each, chunks in order)   whatever the client does there -- launches, waits for the device;
 *                                   the LAST stage may describe one more loop (the assembling of the chunk's results), posted likewise;
 *   finish   (whichever thread ran that loop's last piece)          the chunk's accounting; its buffer set is free again.
 * The stage functions return < 0 on an error (mz_last_error() set): the call is aborted, everything in flight is drained. */
typedef struct mz_flow {
    /* the client's */
    mz_ctx *X;
    void *self;
    int nstage, threaded;
    int (*cut)(void *self, int k, int set, mz_ajob *pack);                            /* 1: chunk k laid out; 0: nothing left; < 0: error */
    int (*stage[MZ_FLOW_STAGES])(void *self, int k, int set, mz_ajob *post);         /* post: NULL except for the last stage */
    int (*finish)(void *self, int k, int set);                                        /* pairs of the chunk without a result (>= 0); < 0: error */
    /* the engine's */
    pthread_mutex_t mu;
    pthread_cond_t cv;
    int packed[MZ_SETS], finished[MZ_SETS];          /* chunk index + 1 of the last chunk packed / finished in the set */
    int through[MZ_FLOW_STAGES + 1];                 /* chunks through stage s, in order */
    int total, rc, left, jobs_out, failed, chunks, hedged;     /* hedged: pieces of the posted loops that were handed out twice */
    mz_ajob pack[MZ_SETS], post[MZ_SETS];
    struct mz_flow_arg { struct mz_flow *F; int k; } parg[MZ_SETS], qarg[MZ_SETS], sarg[MZ_FLOW_STAGES];
    int wait_k;                                      /* what the calling thread is waiting for (flow_ready) */
    char err[600];
} mz_flow;
MZ_INTERNAL int mzi_flow_run(mz_flow *F);            /* pairs without a result, or -1 (mz_last_error()) */

/* dwords of prep records (mz_dev_batch.prep) a pair of N columns can need, whatever kernel the plan gives it: the transposed band of
 * a MZ_MODE_COL pair, 2 (N + 1), in whole 64-dword groups (kernels/plan.inc: szPrep) -- so the chunk pipelines reserve the prep buffer
 * before the plan has run and make the records right behind it */
#define MZ_PREP_BOUND(N) ((2 * ((size_t)(N) + 1) + 63) & ~(size_t)63)

static inline size_t mzi_al256(size_t x) { return (x + 255) & ~(size_t)255; }
static inline double mzi_now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

#endif
