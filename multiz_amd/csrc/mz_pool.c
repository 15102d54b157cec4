/* mz_pool.c -- the host threads of the batch pipeline (mz_batch.c): one process-wide pool of sleeping workers, a
 * parallel-for that several threads may call at once, and -- round 5 -- loops that are POSTED and not waited for
 * (mzi_post): the chunk pipeline (mz_flow.c) posts the packing of chunk k+1 while chunk k's is still being worked on and
 * the assembling of chunk k-3 beside both, so the workers go from piece to piece without a barrier per chunk, and the
 * calling thread works with them (mzi_help_until) instead of waiting.
 *
 * Why not OpenMP here: a libgomp team spins after every parallel region (GOMP_SPINCOUNT), two or three teams of 24
 * spinning threads burn a container's CPU quota in tens of milliseconds (the GPU boxes of this project give a job
 * 16 CPUs' worth of time per 100 ms: cpu.max "1600000 100000"), and the kernel then parks every thread of the process,
 * the ones feeding the GPU included -- mz_yama_batch() of 50 000 pairs took 14 to 52 ms from call to call.  These
 * workers sleep on a condition variable between jobs; what they cost is what they copy.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "mz_ctx.h"

/* (a job is an mz_ajob: mz_ctx.h.  `done` == NULL: somebody waits for it in mzi_parallel_for(); else the thread that
 * finishes its last piece calls done(arg), outside the pool's lock.  A job leaves the list when its last piece is HANDED
 * OUT, so the list only ever holds jobs that still have pieces) */
typedef mz_ajob pjob;

#define POOL_MAX 64
static struct {
    pthread_mutex_t mu;
    pthread_cond_t work, done;
    pjob *head;
    int nthreads, started, quit, cv_ready;
    pthread_t th[POOL_MAX];
} g_pool = { PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, NULL, 0, 0, 0, 0, { 0 } };

/* `work` is also waited on with a deadline (idle_wait): on the MONOTONIC clock -- a step of the wall clock would stretch or swallow the
 * poll -- which a condition variable has to be told when it is made.  (Pool lock held; before anybody has waited on it.) */
static void pool_cv_locked(void)
{
    pthread_condattr_t a;
    if (g_pool.cv_ready) return;
    pthread_condattr_init(&a);
    pthread_condattr_setclock(&a, CLOCK_MONOTONIC);
    pthread_cond_init(&g_pool.work, &a);
    pthread_condattr_destroy(&a);
    g_pool.cv_ready = 1;
}

/* CPUs this process may use at once: the affinity mask, capped by the cgroup's CPU quota (v2 cpu.max, v1 cfs_quota) */
int mzi_cpu_budget(void)
{
    cpu_set_t set;
    int cpus = 0;
    FILE *f;
    if (sched_getaffinity(0, sizeof set, &set) == 0) cpus = CPU_COUNT(&set);
    if (cpus <= 0) cpus = (int)sysconf(_SC_NPROCESSORS_ONLN);
    if (cpus <= 0) cpus = 1;
    if ((f = fopen("/sys/fs/cgroup/cpu.max", "r")) != NULL) {
        char q[64];
        long period = 0;
        if (fscanf(f, "%63s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
            const long quota = atol(q);
            if (quota > 0 && (quota + period - 1) / period < cpus) cpus = (int)((quota + period - 1) / period);
        }
        fclose(f);
    } else if ((f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) != NULL) {
        long quota = -1, period = 100000;
        FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
        if (fscanf(f, "%ld", &quota) != 1) quota = -1;
        if (g) { if (fscanf(g, "%ld", &period) != 1) period = 100000; fclose(g); }
        fclose(f);
        if (quota > 0 && period > 0 && (quota + period - 1) / period < cpus) cpus = (int)((quota + period - 1) / period);
    }
    return cpus;
}

/* ---- pieces, and pieces run twice.
 * A posted loop whose pieces may be run more than once (job->hedge: every piece writes the same bytes whoever runs it, and adds
 * nothing up) is watched after its last piece has been handed out: a piece that has been running for MZ_HEDGE_US (default 400 us --
 * several times what a piece takes) is handed out AGAIN to a thread that has nothing else to do, and the loop is complete when every
 * piece has been finished by somebody.  Why: on the GPU boxes of this project a host thread is now and then off the CPU for 1-6 ms
 * in the middle of a piece (tests/tools/stall_hunt.py: one call in eight of mz_yama_batch() took 1.2-2.4 x the median, and in every
 * one of them ONE piece of a chunk's packing or assembling came in late -- the chunk, and every chunk behind it, waited).  The thread
 * that was late still finishes its piece, some time: whoever owns what the pieces write waits for the loop to be QUIET
 * (mzi_job_quiet) before that is reused or handed on. */
static int g_hedge_us = -1;
static int g_watchers;                                   /* idle threads that look for late pieces (idle_wait) */
static int g_dups, g_dups_max = -1, g_late_first = -1;   /* second runs in progress; at most so many at a time (MZ_HEDGE_DUPS, default 2); MZ_HEDGE_FIRST=0: only idle threads take them */
static pjob *g_out;                                      /* hedged loops with pieces running, all handed out (linked by olink) */

static double now_us(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return 1e6 * t.tv_sec + 1e-3 * t.tv_nsec; }

/* the next piece of the oldest job that has one (pool lock held); the job leaves the list with its last piece */
static pjob *grab_any(int *lo, int *hi, int *piece)
{
    pjob *j = g_pool.head;
    if (!j) return NULL;
    *lo = j->next;
    *hi = j->next + j->grain < j->n ? j->next + j->grain : j->n;
    *piece = j->next / j->grain;
    j->next = *hi;
    j->active++;
    if (j->hedge) { j->state[*piece] = 1; j->t_start[*piece] = now_us(); }
    if (j->next >= j->n) {
        g_pool.head = j->link; j->link = NULL;
        if (j->hedge) {
            j->olink = g_out; g_out = j;
            if (g_watchers < 2) { pthread_cond_signal(&g_pool.work); pthread_cond_signal(&g_pool.work); }     /* (sleepers become the watchers: idle_wait) */
        }
    }
    return j;
}

/* a piece somebody has been at for too long, to be run once more (pool lock held) */
static pjob *grab_late(int *lo, int *hi, int *piece)
{
    pjob *j;
    const double t = now_us();
    /* a box under load makes MANY pieces late at once, and every thread running somebody's piece again instead of a new one makes all of
     * them later still: a few second runs at a time */
    if (g_dups >= g_dups_max) return NULL;
    for (j = g_out; j; j = j->olink) {
        int i;
        /* late: out for longer than MZ_HEDGE_US AND than three times what this loop's pieces have taken so far (a loop of heavy pieces --
         * ten-row blocks: 0.4 ms a piece -- is not late at 0.4 ms) */
        const double avg3 = j->n_done ? 3.0 * j->t_sum / j->n_done : 4.0 * g_hedge_us, late = avg3 > g_hedge_us ? avg3 : g_hedge_us;
        double oldest = t;
        /* Every thread comes by here before it takes a piece (late pieces first), under the pool's lock: a walk over all pieces of all
         * loops that are out, a hundred times per loop, cost a quiet box up to 9 % of a call of heavy pieces (C3: 4.93 against 4.53 ms with
         * second runs off, profiles/r6_stall_hunt.txt).  A loop whose oldest piece out is younger than the shortest "late" there is is
         * passed by until that piece could be late. */
        if (t < j->skip_until) continue;
        for (i = 0; i < j->npiece; ++i)
            if (j->state[i] == 1) {
                if (t - j->t_start[i] > late) {
                    j->state[i] = 3;                     /* (twice is enough) */
                    j->active++;
                    j->hedged++;
                    g_dups++;
                    *piece = i; *lo = i * j->grain; *hi = *lo + j->grain < j->n ? *lo + j->grain : j->n;
                    return j;
                }
                if (j->t_start[i] < oldest) oldest = j->t_start[i];
            }
        j->skip_until = oldest + g_hedge_us;             /* (pieces handed out from now on are younger still) */
    }
    return NULL;
}

/* one run of piece `piece` = [lo, hi) of j has ended (pool lock held on entry and on return; released around a posted job's callback) */
static void piece_done(pjob *j, int lo, int hi, int piece)
{
    int complete = 0;
    j->active--;
    if (!j->hedge) { j->pending -= hi - lo; complete = j->pending == 0; }
    else if (j->state[piece] != 2) { j->state[piece] = 2; j->pending -= hi - lo; complete = j->pending == 0; j->t_sum += now_us() - j->t_start[piece]; j->n_done++; }
    if (complete && j->hedge) { pjob **pp; for (pp = &g_out; *pp && *pp != j; pp = &(*pp)->olink) ; if (*pp) *pp = j->olink; j->olink = NULL; }
    if (j->active == 0 || (complete && !j->done)) pthread_cond_broadcast(&g_pool.done);      /* (somebody may wait for the loop to be quiet) */
    if (complete && j->done) {
        void (*done)(void *) = j->done;
        void *arg = j->arg;
        pthread_mutex_unlock(&g_pool.mu);                /* (the callback may re-post the job -- once it is quiet) */
        done(arg);
        pthread_mutex_lock(&g_pool.mu);
    }
}

/* nothing to do (pool lock held): sleep until something is posted -- or, while hedged loops have pieces out, for a fraction of the
 * hedging time */
/* Nothing to take.  While hedged loops are out, TWO of the idle threads look again after half the hedge time (a piece may have
 * become late meanwhile); the others sleep until there is new work -- with every idle thread polling, 24 threads woke 5 000 times
 * a second each during a call and a quiet box's calls took 2 % longer than without hedging (C2: 8.41 against 8.26 ms). */
static void idle_wait(void)
{
    if (g_out && g_watchers < 2) {
        struct timespec t;
        clock_gettime(CLOCK_MONOTONIC, &t);                  /* (the clock `work` was made for: pool_cv_locked) */
        t.tv_nsec += 1000L * (g_hedge_us / 2 > 50 ? g_hedge_us / 2 : 50);
        if (t.tv_nsec >= 1000000000L) { t.tv_nsec -= 1000000000L; t.tv_sec++; }
        ++g_watchers;
        pthread_cond_timedwait(&g_pool.work, &g_pool.mu, &t);
        --g_watchers;
    } else pthread_cond_wait(&g_pool.work, &g_pool.mu);
}

/* MZ_HEDGE_DELAY_US (tests): a piece's SECOND run starts that much later -- long after the first one is through and the chunk's next stages
 * are at work on what the piece wrote and read: what a second run must not disturb (tests/test_gpu_parity.py) */
static int g_hedge_delay_us;                              /* (read when the pool starts, under its lock) */
static void late_run_delay(void)
{
    const int us = g_hedge_delay_us;
    if (us) { struct timespec t = { us / 1000000, 1000L * (us % 1000000) }; nanosleep(&t, NULL); }
}

static void *pool_worker(void *arg)
{
    (void)arg;
    pthread_mutex_lock(&g_pool.mu);
    while (!g_pool.quit) {
        pjob *j;
        int lo, hi, piece;
        int late = 0;
        /* a late piece first: it belongs to an older loop than any piece not yet handed out, and that loop's chunk is what the GPU waits for
         * (late pieces used to be looked for only by threads with nothing else to take: with three chunks' loops queued a stalled piece of
         * the first waited 3 ms for its second run) */
        if (!(g_late_first && g_out && (j = grab_late(&lo, &hi, &piece)) && (late = 1)) && !(j = grab_any(&lo, &hi, &piece)) &&
            !((j = grab_late(&lo, &hi, &piece)) && (late = 1))) { idle_wait(); continue; }
        pthread_mutex_unlock(&g_pool.mu);
        if (late) late_run_delay();
        j->fn(j->ctx, lo, hi);
        pthread_mutex_lock(&g_pool.mu);
        if (late) g_dups--;
        piece_done(j, lo, hi, piece);
    }
    pthread_mutex_unlock(&g_pool.mu);
    return NULL;
}

/* workers: MZ_HOST_THREADS, or MZ_COPY_THREADS (fewer on a machine with fewer CPUs) -- more than a CPU quota's worth when
 * there is one: the calls come in bursts (a quota of 16 CPUs is 1.6 s of CPU time per 100 ms, a 50 000-pair call spends
 * 0.15 s of it), and measured on such a box 24 threads beat 16 by a fifth */
static void pool_start_locked(void)
{
    const char *e = getenv("MZ_HOST_THREADS");
    int want = e && atoi(e) > 0 ? atoi(e) : (int)sysconf(_SC_NPROCESSORS_ONLN), i;
    pool_cv_locked();
    if (want > MZ_COPY_THREADS && !(e && atoi(e) > 0)) want = MZ_COPY_THREADS;
    if (want < 1) want = 1;
    if (want > POOL_MAX) want = POOL_MAX;
    if (g_hedge_us < 0) { const char *h = getenv("MZ_HEDGE_US"); g_hedge_us = h ? atoi(h) : 400; }     /* (0: pieces are never run twice) */
    { const char *h = getenv("MZ_HEDGE_DELAY_US"); g_hedge_delay_us = h && atoi(h) > 0 ? atoi(h) : 0; }
    if (g_dups_max < 0) { const char *h = getenv("MZ_HEDGE_DUPS"); g_dups_max = h && atoi(h) > 0 ? atoi(h) : 2; }
    if (g_late_first < 0) { const char *h = getenv("MZ_HEDGE_FIRST"); g_late_first = !(h && h[0] == '0'); }
    g_pool.started = 1;
    g_pool.nthreads = 0;
    for (i = 0; i < want - 1; ++i) {
        if (pthread_create(&g_pool.th[g_pool.nthreads], NULL, pool_worker, NULL) != 0) break;
        ++g_pool.nthreads;
    }
}

int mzi_pool_threads(void)
{
    int n;
    pthread_mutex_lock(&g_pool.mu);
    if (!g_pool.started) pool_start_locked();
    n = g_pool.nthreads + 1;
    pthread_mutex_unlock(&g_pool.mu);
    return n;
}

static void enqueue(pjob *job)                            /* (pool lock held) jobs in arrival order: the older chunk first */
{
    pjob **pp;
    job->next = 0; job->pending = job->n; job->link = NULL; job->olink = NULL; job->active = 0; job->hedged = 0; job->t_sum = 0; job->n_done = 0; job->skip_until = 0;
    if (!g_pool.started) pool_start_locked();
    job->npiece = (job->n + job->grain - 1) / job->grain;
    if (job->hedge && (!job->done || g_hedge_us <= 0 || job->npiece > MZ_HEDGE_PIECES)) job->hedge = 0;
    if (job->hedge) memset(job->state, 0, (size_t)job->npiece);
    for (pp = &g_pool.head; *pp; pp = &(*pp)->link) ;
    *pp = job;
    pthread_cond_broadcast(&g_pool.work);
}

/* fn(ctx, lo, hi) over [0, n) in pieces of `grain`, on the pool's workers and the calling thread; returns when every
 * piece is done.  Small loops run in the caller alone. */
void mzi_parallel_for(int n, int grain, mz_pfn fn, void *ctx)
{
    pjob job;
    if (n <= 0) return;
    if (grain < 1) grain = 1;
    if (n <= grain) { fn(ctx, 0, n); return; }
    memset(&job, 0, sizeof job);
    job.fn = fn; job.ctx = ctx; job.n = n; job.grain = grain;
    pthread_mutex_lock(&g_pool.mu);
    enqueue(&job);
    while (job.next < job.n) {                               /* the caller works on ITS loop (older jobs are the workers') */
        const int lo = job.next, hi = lo + grain < n ? lo + grain : n;
        job.next = hi;
        job.active++;
        if (hi >= n) { pjob **pp; for (pp = &g_pool.head; *pp != &job; pp = &(*pp)->link) ; *pp = job.link; job.link = NULL; }
        pthread_mutex_unlock(&g_pool.mu);
        fn(ctx, lo, hi);
        pthread_mutex_lock(&g_pool.mu);
        job.active--;
        job.pending -= hi - lo;
    }
    while (job.pending > 0 || job.active > 0) pthread_cond_wait(&g_pool.done, &g_pool.mu);
    pthread_mutex_unlock(&g_pool.mu);
}

/* The same loop, not waited for: returns at once; job->done(job->arg) is called -- by whichever thread finishes the last
 * piece -- when all of [0, n) has been run.  The job must stay where it is until then.  n == 0: done() is called here. */
void mzi_post(mz_ajob *job)
{
    if (job->grain < 1) job->grain = 1;
    job->active = 0; job->hedged = 0;
    if (job->n <= 0) { if (job->done) job->done(job->arg); return; }
    pthread_mutex_lock(&g_pool.mu);
    enqueue(job);
    pthread_mutex_unlock(&g_pool.mu);
}

/* The calling thread works on posted pieces until ready(arg) says so.  `ready` is evaluated under the pool's lock and
 * must not take any other; whoever makes it true calls mzi_pool_kick() afterwards, so a helper asleep for want of
 * pieces is woken. */
void mzi_help_until(int (*ready)(void *), void *arg)
{
    pthread_mutex_lock(&g_pool.mu);
    pool_cv_locked();
    while (!ready(arg)) {
        pjob *j;
        int lo, hi, piece;
        int late = 0;
        /* a late piece first: it belongs to an older loop than any piece not yet handed out, and that loop's chunk is what the GPU waits for
         * (late pieces used to be looked for only by threads with nothing else to take: with three chunks' loops queued a stalled piece of
         * the first waited 3 ms for its second run) */
        if (!(g_late_first && g_out && (j = grab_late(&lo, &hi, &piece)) && (late = 1)) && !(j = grab_any(&lo, &hi, &piece)) &&
            !((j = grab_late(&lo, &hi, &piece)) && (late = 1))) { idle_wait(); continue; }
        pthread_mutex_unlock(&g_pool.mu);
        if (late) late_run_delay();
        j->fn(j->ctx, lo, hi);
        pthread_mutex_lock(&g_pool.mu);
        if (late) g_dups--;
        piece_done(j, lo, hi, piece);
    }
    pthread_mutex_unlock(&g_pool.mu);
}
/* wait until nobody is running a piece of a posted loop any more (it is complete: its callback has been called); returns how many
 * of its pieces were handed out twice */
int mzi_job_quiet(mz_ajob *job)
{
    int h;
    pthread_mutex_lock(&g_pool.mu);
    while (job->active > 0) pthread_cond_wait(&g_pool.done, &g_pool.mu);
    h = job->hedged;
    pthread_mutex_unlock(&g_pool.mu);
    return h;
}
void mzi_pool_kick(void)
{
    pthread_mutex_lock(&g_pool.mu);
    pool_cv_locked();
    pthread_cond_broadcast(&g_pool.work);
    pthread_mutex_unlock(&g_pool.mu);
}

/* mz_finalize(): the workers go (a later call starts new ones) */
void mzi_pool_stop(void)
{
    int i, n;
    pthread_mutex_lock(&g_pool.mu);
    if (!g_pool.started || g_pool.head) { pthread_mutex_unlock(&g_pool.mu); return; }
    g_pool.quit = 1;
    n = g_pool.nthreads;
    pthread_cond_broadcast(&g_pool.work);
    pthread_mutex_unlock(&g_pool.mu);
    for (i = 0; i < n; ++i) pthread_join(g_pool.th[i], NULL);
    pthread_mutex_lock(&g_pool.mu);
    g_pool.quit = 0; g_pool.started = 0; g_pool.nthreads = 0;
    pthread_mutex_unlock(&g_pool.mu);
}

/* ------------------------------------------------------------------------------------------------ stage threads
 * The helper threads of a context's chunk pipelines (mz_batch.c: launcher, collector; mz_prebatch.c: two launchers, collector):
 * persistent, asleep between calls.  fn(job) runs on the worker; the worker counts as free when fn has returned. */
static void *stage_worker(void *arg)
{
    mz_worker *w = (mz_worker *)arg;
    for (;;) {
        void (*fn)(void *);
        void *job;
        pthread_mutex_lock(&w->mu);
        while (!w->quit && !w->job) pthread_cond_wait(&w->cv, &w->mu);
        if (w->quit) { pthread_mutex_unlock(&w->mu); break; }
        fn = w->fn; job = w->job;
        pthread_mutex_unlock(&w->mu);
        fn(job);
        pthread_mutex_lock(&w->mu);
        w->job = NULL;
        pthread_cond_broadcast(&w->cv);                  /* (a giver may be waiting for the worker to come free) */
        pthread_mutex_unlock(&w->mu);
    }
    return NULL;
}

int mzi_workers_start(mz_worker *w, int n)
{
    int i;
    for (i = 0; i < n; ++i) {
        if (w[i].started) continue;
        pthread_mutex_init(&w[i].mu, NULL);
        pthread_cond_init(&w[i].cv, NULL);
        w[i].quit = 0; w[i].job = NULL;
        if (pthread_create(&w[i].th, NULL, stage_worker, &w[i]) != 0) { pthread_mutex_destroy(&w[i].mu); pthread_cond_destroy(&w[i].cv); return -1; }
        w[i].started = 1;
    }
    return 0;
}

void mzi_worker_give(mz_worker *w, void (*fn)(void *), void *job)
{
    pthread_mutex_lock(&w->mu);
    while (w->job) pthread_cond_wait(&w->cv, &w->mu);    /* the previous call's stage has reported its end but not yet returned */
    w->fn = fn; w->job = job;
    pthread_cond_broadcast(&w->cv);
    pthread_mutex_unlock(&w->mu);
}

void mzi_workers_end(mz_worker *w, int n)
{
    int i;
    for (i = 0; i < n; ++i) {
        if (!w[i].started) continue;
        pthread_mutex_lock(&w[i].mu);
        w[i].quit = 1;
        pthread_cond_broadcast(&w[i].cv);
        pthread_mutex_unlock(&w[i].mu);
        pthread_join(w[i].th, NULL);
        pthread_mutex_destroy(&w[i].mu);
        pthread_cond_destroy(&w[i].cv);
        w[i].started = 0;
    }
}

/* ------------------------------------------------------------------------------------------------ result blocks
 * The merged columns of a chunk are ONE malloc()ed block (mz_out.block).  A 27 MB block is an mmap() of its own
 * to malloc: fresh zero pages on every call (52 000 page faults per 50 000-pair C2 call, 600 000 for a C4 share) and an
 * munmap() on free.  mz_free_outs() therefore parks up to BLOCK_KEEP blocks / BLOCK_KEEP_BYTES here and the next calls
 * take them back, warm; they stay ordinary malloc() pointers (a caller that free()s one itself just does not
 * return it). */
#include <malloc.h>
#define BLOCK_KEEP 32
#define BLOCK_KEEP_BYTES ((size_t)3 << 30)
static struct { pthread_mutex_t mu; void *p[BLOCK_KEEP]; size_t cap[BLOCK_KEEP]; size_t bytes; int n; } g_blocks = { PTHREAD_MUTEX_INITIALIZER, { 0 }, { 0 }, 0, 0 };

void *mzi_block_get(size_t need)
{
    void *p = NULL;
    int i, best = -1;
    pthread_mutex_lock(&g_blocks.mu);
    for (i = 0; i < g_blocks.n; ++i)                         /* the smallest that fits, if it is not absurdly large for the job */
        if (g_blocks.cap[i] >= need && (best < 0 || g_blocks.cap[i] < g_blocks.cap[best])) best = i;
    if (best >= 0 && g_blocks.cap[best] <= 2 * need + (1 << 20)) {
        p = g_blocks.p[best];
        g_blocks.bytes -= g_blocks.cap[best];
        g_blocks.p[best] = g_blocks.p[--g_blocks.n]; g_blocks.cap[best] = g_blocks.cap[g_blocks.n];
    }
    pthread_mutex_unlock(&g_blocks.mu);
    return p ? p : malloc(need);
}

void mzi_block_put(void *p)
{
    size_t cap;
    if (!p) return;
    cap = malloc_usable_size(p);
    pthread_mutex_lock(&g_blocks.mu);
    if (cap >= ((size_t)1 << 20) && g_blocks.n < BLOCK_KEEP && g_blocks.bytes + cap <= BLOCK_KEEP_BYTES) {
        g_blocks.p[g_blocks.n] = p; g_blocks.cap[g_blocks.n++] = cap; g_blocks.bytes += cap;
        p = NULL;
    }
    pthread_mutex_unlock(&g_blocks.mu);
    free(p);
}

void mzi_blocks_drop(void)
{
    int i;
    pthread_mutex_lock(&g_blocks.mu);
    for (i = 0; i < g_blocks.n; ++i) free(g_blocks.p[i]);
    g_blocks.n = 0; g_blocks.bytes = 0;
    pthread_mutex_unlock(&g_blocks.mu);
}
