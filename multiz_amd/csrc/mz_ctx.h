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
    mz_worker pworker[3];                  /* first launcher, second launcher, collector of mz_preyama_batch() */
    hipStream_t bstream[MZ_SETS];          /* one stream per set (mz_yama_batch) */
    hipStream_t ustream[MZ_SETS];          /* and one of high priority for its upload, expansion and plan: a chunk's plan must not queue
                                            * behind the DP waves of the chunks before it (the launcher waits for its totals) */
    hipEvent_t bdone[MZ_SETS], bplan[MZ_SETS];
    hipEvent_t btime[MZ_SETS][6];          /* MZ_TIMING=2: start, uploaded, planned, DP done, results packed, copied back */
    int btime_ready;
    struct { const void *key; hipEvent_t done; int used; } ws[MZ_WS_MAX];
    int ws_victim;
    int copy_threads;                      /* host threads of this context's pack / unpack loops */
    mz_worker worker[2];                   /* launcher, collector (mz_batch.c) */
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
MZ_INTERNAL int mzi_ensure_init(void);
extern MZ_INTERNAL __thread int mzi_warm_thread;     /* set in the thread of mz_warm_start(): its batch prints no MZ_TIMING line */
MZ_INTERNAL int mzi_sync_scores(void);
MZ_INTERNAL int mzi_timing(void);                       /* MZ_TIMING, parsed once: 0 quiet, 1 per call, 2 per chunk */
MZ_INTERNAL void mzi_workers_stop(mz_ctx *X);           /* mz_batch.c: ctx_close() ends the context's helper threads */
MZ_INTERNAL int mzi_deal_snake(int n, const double *weight, int use, int *owner, int *where, int *cnt, int *start);    /* mz_batch.c */
MZ_INTERNAL int mzi_pre_on_ctx(mz_ctx *X, int n, const mz_prejob *jobs, mz_preout *outs, int64_t stats[3]);   /* mz_prebatch.c */

/* mz_pool.c: the host threads of the batch pipeline (no OpenMP there: see the file), recycled result blocks */
typedef void (*mz_pfn)(void *ctx, int lo, int hi);
MZ_INTERNAL void mzi_parallel_for(int n, int grain, mz_pfn fn, void *ctx);
MZ_INTERNAL int mzi_pool_threads(void);
MZ_INTERNAL int mzi_cpu_budget(void);
MZ_INTERNAL void mzi_pool_stop(void);
MZ_INTERNAL void *mzi_block_get(size_t need);
MZ_INTERNAL void mzi_block_put(void *p);
MZ_INTERNAL void mzi_blocks_drop(void);

static inline size_t mzi_al256(size_t x) { return (x + 255) & ~(size_t)255; }
static inline double mzi_now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

#endif
