/* mz_flow.c -- the chunk pipeline shared by mz_yama_batch() (mz_batch.c) and mz_preyama_batch() (mz_prebatch.c); the
 * interface is in mz_ctx.h.
 *
 * Round 4's pipelines had the calling thread pack a chunk (a parallel loop it waited for), then issue that chunk's
 * upload and plan, then pack the next: the GPU got a chunk every 0.66-0.75 ms whose DP takes 0.4 ms of its time, thirteen
 * DPs ran strictly one after another (profiles/r4_timeline_host_c2.txt) and the pool's threads stood at a barrier twice
 * per chunk.  Here nothing on the host waits for a loop: packing and assembling are posted to the pool piece by piece in
 * chunk order, the calling thread works there too, and issuing is the stage threads' business alone.
 */
#include <stdio.h>
#include <string.h>

#include "mz_ctx.h"

/* The chunk streams (mz_ctx.h), and which hardware queue each of them gets.
 *
 * Measured on MI355X (tests/tools/ub/chain.hip; profiles/r5_pipes.txt): the runtime gives every new stream the next hardware queue
 * (until GPU_MAX_HW_QUEUES exist; later streams share), hardware queue i belongs to pipe i mod 4 of the GPU's command processor, and
 * a queue whose pipe-mate has a kernel RUNNING gets each of its own packets started 60-200 us late -- however small the kernel, however
 * much room the GPU has (a chain of tiny kernels: 3 us per kernel alone or beside kernels that fill the GPU from other pipes, 60-210 us
 * beside one on the same pipe).  A chunk's front is a chain of a dozen small dependent kernels; with the streams as they came (round 4,
 * and at first in round 5) it took ~1 ms per chunk beside the DPs, the plan of chunk k+2 was not through when the DP of chunk k ended,
 * and the DP streams idled a quarter of the time.  So the streams are created in an order that puts on one pipe what can wait for each
 * other:
 *        pipe a            pipe b            pipe c              pipe d
 *        DP stream 0       DP stream 1       the fronts          the link copies (host -> device)
 *        tail stream 0     tail stream 1
 *        lane of DP 0      lane of DP 1                          (created when the first chunk with several kinds of pairs comes by;
 *        2nd lane          2nd lane                               two filler streams complete every round of four)
 * -- a front's short kernels wait for nothing but each other; the copies, which bound the fronts' rate (a chunk's 13 MB at the link's
 * ~50 GB/s), wait for nothing at all; a tail's kernels (walk, script packing, results -> host: latency-bound anyway) start late beside
 * a running DP and may in turn start a DP 0.1 ms late, which the other DP stream's blocks cover.  This needs the hardware queues in a
 * row: the library asks the runtime for 24 (init_devices()).  If the runtime hands them out differently nothing breaks; the fronts
 * are slow again.  MZ_TAILS=1, MZ_LANES=0..2 (default 2): fewer streams (measurements). */
static int flow_new_stream(mz_ctx *X, hipStream_t *s)
{
    if (X->nqall >= MZ_QALL) return mzi_set_err("chunk pipeline: out of stream slots");
    *s = NULL;
    if (mzi_lazy_stream(s)) return -1;
    X->qall[X->nqall++] = *s;
    return 0;
}

int mzi_flow_streams(mz_ctx *X)
{
    hipStream_t filler;
    int i, nt;
    const double t0 = mzi_now_s();
    if (X->nq) return 0;
    { const char *e = getenv("MZ_TAILS"); nt = e && atoi(e) == 1 ? 1 : 2; }
    /* round 0: DP 0, DP 1, fronts, copies; round 1: tail 0, tail 1 (the lanes' rounds: mzi_flow_lanes) */
    if (flow_new_stream(X, &X->qd[0]) || flow_new_stream(X, &X->qd[1]) || flow_new_stream(X, &X->qf[0]) || flow_new_stream(X, &X->qc)) return -1;
    {
        /* MZ_TAIL_PIPE (measurements): 0 = a tail beside ITS chunk's DP stream (pipes a, b); 1 = beside the OTHER DP stream (chunk k's walk
         * starts when DP k ends -- and so should DP k+2, on the same queue as DP k: with the tail on that pipe its packets come 60-200 us
         * late each while the walk runs); 2 = both tails on the copies' pipe d; 3 = both on the fronts' pipe c */
        static int tp = -1;
        if (tp < 0) { const char *e = getenv("MZ_TAIL_PIPE"); tp = e ? atoi(e) : 0; if (tp < 0 || tp > 3) tp = 0; }
        if (tp <= 1) {
            hipStream_t a, b;
            if (flow_new_stream(X, &a) || flow_new_stream(X, nt > 1 ? &b : &filler)) return -1;
            if (nt > 1) { X->qt[0] = tp ? b : a; X->qt[1] = tp ? a : b; } else X->qt[0] = a;
        } else {
            for (i = 0; i < nt; ++i) {
                while (X->nqall % 4 != (tp == 2 ? 3 : 2)) if (flow_new_stream(X, &filler)) return -1;
                if (flow_new_stream(X, &X->qt[i])) return -1;
            }
        }
    }
    for (i = 0; i < 2; ++i) {
        X->qlane[i].n = 0;
        HIPCK(hipEventCreateWithFlags((hipEvent_t *)&X->qlane[i].fork, mzi_event_flags()));
    }
    X->nf = 1; X->nt = nt; X->nq = 2;
    if (mzi_timing()) fprintf(stderr, "{\"mz_flow_streams\": {\"created\": %d, \"ms\": %.1f}}\n", X->nqall, 1e3 * (mzi_now_s() - t0));
    return 0;
}

/* Calls of few LONG pairs (BASELINE config 5: 1 000 pairs of 100 000 columns, four chunks of 250): a chunk's DP is a wave per pair and
 * takes as long as ONE pair takes -- 25-30 ms -- however few it holds, the four of them fill a quarter of the GPU's SIMDs each, and two
 * DP streams ran them two after two: 56 ms a call.  Four DP streams (and a tail each) run them side by side.  Which pipe these streams
 * land on does not matter here: their kernels take milliseconds. */
int mzi_flow_wide(mz_ctx *X)
{
    int i;
    if (X->nq_wide) return 0;
    for (i = 2; i < MZ_QS; ++i) {
        if (flow_new_stream(X, &X->qd[i]) || flow_new_stream(X, &X->qt[i])) return -1;
        X->qlane[i].n = 0;
        HIPCK(hipEventCreateWithFlags((hipEvent_t *)&X->qlane[i].fork, mzi_event_flags()));
    }
    X->nq_wide = MZ_QS;
    return 0;
}

/* the DP streams' lanes, when the first chunk with several kinds of pairs needs them (from a launcher's thread; ~5 ms a stream).  The text
 * path has TWO launch stages, each on a thread of its own: one at a time in here (an intermittent segmentation fault of
 * `bench.py --config c4i` -- mixed kinds, two-stage merges -- was two threads creating the lanes at once). */
static pthread_mutex_t g_lanes_mu = PTHREAD_MUTEX_INITIALIZER;
static int flow_lanes_locked(mz_ctx *X)
{
    static int want = -1;
    hipStream_t filler;
    int i, l;
    if (want < 0) { const char *e = getenv("MZ_LANES"); want = e && atoi(e) >= 0 ? atoi(e) : 2; if (want > 2) want = 2; }
    if (X->lanes_made || !want) { X->lanes_made = 1; return 0; }
    for (l = 0; l < want; ++l) {
        while (X->nqall % 4) if (flow_new_stream(X, &filler)) return -1;        /* the next stream: pipe a */
        for (i = 0; i < 2; ++i) {
            hipStream_t s;
            if (flow_new_stream(X, &s)) return -1;
            X->qlane[i].stream[l] = (void *)s;
            HIPCK(hipEventCreateWithFlags((hipEvent_t *)&X->qlane[i].join[l], mzi_event_flags()));
        }
    }
    for (i = 0; i < 2; ++i) X->qlane[i].n = want;
    __atomic_store_n(&X->lanes_made, 1, __ATOMIC_RELEASE);
    return 0;
}
int mzi_flow_lanes(mz_ctx *X)
{
    int rc;
    pthread_mutex_lock(&g_lanes_mu);
    rc = flow_lanes_locked(X);
    pthread_mutex_unlock(&g_lanes_mu);
    return rc;
}

/* everything a context's chunk pipelines may have in flight (after an error) */
void mzi_flow_sync(mz_ctx *X)
{
    int s, l;
    if (X->stream) hipStreamSynchronize(X->stream);
    if (X->qc) hipStreamSynchronize(X->qc);
    for (s = 0; s < X->nf; ++s) hipStreamSynchronize(X->qf[s]);
    for (s = 0; s < X->nt; ++s) hipStreamSynchronize(X->qt[s]);
    for (s = 0; s < (X->nq_wide ? X->nq_wide : X->nq); ++s) {
        hipStreamSynchronize(X->qd[s]);
        if (s >= X->nt && X->qt[s]) hipStreamSynchronize(X->qt[s]);
        for (l = 0; l < X->qlane[s].n; ++l) hipStreamSynchronize((hipStream_t)X->qlane[s].stream[l]);
    }
}

static void flow_abort(mz_flow *F)
{
    pthread_mutex_lock(&F->mu);
    if (F->rc >= 0) { F->rc = -1; snprintf(F->err, sizeof F->err, "%s", mz_last_error()); }
    pthread_cond_broadcast(&F->cv);
    pthread_mutex_unlock(&F->mu);
    mzi_pool_kick();
}

static void flow_pack_done(void *arg)
{
    struct mz_flow_arg *a = (struct mz_flow_arg *)arg;
    mz_flow *F = a->F;
    pthread_mutex_lock(&F->mu);
    F->packed[a->k % MZ_SETS] = a->k + 1;
    F->jobs_out--;
    pthread_cond_broadcast(&F->cv);
    pthread_mutex_unlock(&F->mu);
    mzi_pool_kick();
}

static void flow_post_done(void *arg)
{
    struct mz_flow_arg *a = (struct mz_flow_arg *)arg;
    mz_flow *F = a->F;
    const int k = a->k, r = F->finish(F->self, k, k % MZ_SETS);
    if (r < 0) flow_abort(F);
    pthread_mutex_lock(&F->mu);
    if (r > 0) F->failed += r;
    __atomic_store_n(&F->finished[k % MZ_SETS], k + 1, __ATOMIC_RELEASE);
    F->jobs_out--;
    pthread_cond_broadcast(&F->cv);
    pthread_mutex_unlock(&F->mu);
    mzi_pool_kick();
}

/* stage s = 1 .. nstage: for chunk k = 0, 1, ...: wait until the stage before is through with it, do this one's part */
static void flow_stage(void *arg)
{
    struct mz_flow_arg *a = (struct mz_flow_arg *)arg;
    mz_flow *F = a->F;
    const int s = a->k;
    int k;
    hipSetDevice(F->X->device);
    for (k = 0;; ++k) {
        const int set = k % MZ_SETS, last = s == F->nstage;
        int rc;
        pthread_mutex_lock(&F->mu);
        while (F->rc >= 0 && !(s == 1 ? F->packed[set] == k + 1 : F->through[s - 1] > k) && (F->total < 0 || k < F->total)) pthread_cond_wait(&F->cv, &F->mu);
        if (F->rc < 0 || (F->total >= 0 && k >= F->total)) { pthread_mutex_unlock(&F->mu); break; }
        pthread_mutex_unlock(&F->mu);
        if (last) { memset(&F->post[set], 0, sizeof F->post[set]); F->qarg[set].F = F; F->qarg[set].k = k; }
        rc = F->stage[s - 1](F->self, k, set, last ? &F->post[set] : NULL);
        if (rc < 0) { flow_abort(F); break; }
        pthread_mutex_lock(&F->mu);
        F->through[s] = k + 1;
        if (last) F->jobs_out++;
        pthread_cond_broadcast(&F->cv);
        pthread_mutex_unlock(&F->mu);
        if (last) { F->post[set].done = flow_post_done; F->post[set].arg = &F->qarg[set]; mzi_post(&F->post[set]); }
    }
    pthread_mutex_lock(&F->mu);                          /* the last thing a stage does with the flow */
    F->left++;
    pthread_cond_broadcast(&F->cv);
    pthread_mutex_unlock(&F->mu);
    mzi_pool_kick();
}

/* what the calling thread waits for while it works in the pool (evaluated under the pool's lock: atomics only) */
static int flow_set_free(void *arg)
{
    mz_flow *F = (mz_flow *)arg;
    return __atomic_load_n(&F->rc, __ATOMIC_ACQUIRE) < 0 || __atomic_load_n(&F->finished[F->wait_k % MZ_SETS], __ATOMIC_ACQUIRE) == F->wait_k + 1;
}
static int flow_drained(void *arg)
{
    mz_flow *F = (mz_flow *)arg;
    return __atomic_load_n(&F->left, __ATOMIC_ACQUIRE) == F->nstage && __atomic_load_n(&F->jobs_out, __ATOMIC_ACQUIRE) == 0;
}

int mzi_flow_run(mz_flow *F)
{
    int k, s, rc;
    F->total = -1; F->rc = 0; F->left = 0; F->jobs_out = 0; F->failed = 0; F->chunks = 0; F->hedged = 0; F->err[0] = 0;
    memset(F->pack, 0, sizeof F->pack); memset(F->post, 0, sizeof F->post);
    memset(F->packed, 0, sizeof F->packed); memset(F->finished, 0, sizeof F->finished); memset(F->through, 0, sizeof F->through);
    if (F->threaded && (mzi_flow_streams(F->X) || mzi_workers_start(F->X->fworker, F->nstage))) F->threaded = 0;
    if (!F->threaded) {
        /* a chunk at a time, everything here (a single yama() call: no thread is woken) */
        for (k = 0;; ++k) {
            mz_ajob job;
            int r;
            memset(&job, 0, sizeof job);
            if ((r = F->cut(F->self, k, 0, &job)) <= 0) { if (r < 0) return -1; break; }
            mzi_parallel_for(job.n, job.grain, job.fn, job.ctx);
            for (s = 1; s <= F->nstage; ++s) {
                memset(&job, 0, sizeof job);
                if (F->stage[s - 1](F->self, k, 0, s == F->nstage ? &job : NULL) < 0) return -1;
            }
            if (job.fn) mzi_parallel_for(job.n, job.grain, job.fn, job.ctx);
            if ((r = F->finish(F->self, k, 0)) < 0) return -1;
            F->failed += r;
            F->chunks = k + 1;
        }
        return F->failed;
    }
    pthread_mutex_init(&F->mu, NULL);
    pthread_cond_init(&F->cv, NULL);
    for (s = 1; s <= F->nstage; ++s) { F->sarg[s - 1].F = F; F->sarg[s - 1].k = s; mzi_worker_give(&F->X->fworker[s - 1], flow_stage, &F->sarg[s - 1]); }
    for (k = 0;; ++k) {
        const int set = k % MZ_SETS;
        int r;
        if (k >= MZ_SETS) {                                  /* its buffer set is still in use ... */
            F->wait_k = k - MZ_SETS; mzi_help_until(flow_set_free, F);
            F->hedged += mzi_job_quiet(&F->pack[set]) + mzi_job_quiet(&F->post[set]);      /* ... or a thread that came late is still writing to it (mz_pool.c) */
        }
        if (__atomic_load_n(&F->rc, __ATOMIC_ACQUIRE) < 0) break;
        memset(&F->pack[set], 0, sizeof F->pack[set]);
        if ((r = F->cut(F->self, k, set, &F->pack[set])) <= 0) { if (r < 0) flow_abort(F); break; }
        F->parg[set].F = F; F->parg[set].k = k;
        F->pack[set].done = flow_pack_done; F->pack[set].arg = &F->parg[set];
        pthread_mutex_lock(&F->mu);
        F->jobs_out++;
        pthread_mutex_unlock(&F->mu);
        mzi_post(&F->pack[set]);
    }
    pthread_mutex_lock(&F->mu);
    F->total = k; F->chunks = k;
    pthread_cond_broadcast(&F->cv);
    pthread_mutex_unlock(&F->mu);
    mzi_help_until(flow_drained, F);
    /* flow_drained() reads its counters without the flow's lock; the stage thread or pool thread that made it true did so UNDER the lock and
     * still has to broadcast and unlock -- and the flow goes away when this function returns (its memory back to the system, if malloc() had
     * mapped it: a segmentation fault in that thread, one `bench.py --config c4i` in twenty).  Through the lock once: they are out. */
    pthread_mutex_lock(&F->mu);
    pthread_mutex_unlock(&F->mu);
    for (s = 0; s < MZ_SETS && s < k; ++s) F->hedged += mzi_job_quiet(&F->pack[s]) + mzi_job_quiet(&F->post[s]);     /* nobody is writing to the results any more */
    rc = F->rc < 0 ? -1 : F->failed;
    if (rc < 0) mzi_set_err("%s", F->err);
    pthread_mutex_destroy(&F->mu);
    pthread_cond_destroy(&F->cv);
    return rc;
}
