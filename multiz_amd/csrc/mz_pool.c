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
    int nthreads, started, quit;
    pthread_t th[POOL_MAX];
} g_pool = { PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, NULL, 0, 0, 0, { 0 } };

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

/* the next piece of the oldest job that has one (pool lock held); the job leaves the list with its last piece */
static pjob *grab_any(int *lo, int *hi)
{
    pjob *j = g_pool.head;
    if (!j) return NULL;
    *lo = j->next;
    *hi = j->next + j->grain < j->n ? j->next + j->grain : j->n;
    j->next = *hi;
    if (j->next >= j->n) { g_pool.head = j->link; j->link = NULL; }
    return j;
}

/* a piece [lo, hi) of j has been run (pool lock held on entry and on return; released around a posted job's callback) */
static void piece_done(pjob *j, int lo, int hi)
{
    j->pending -= hi - lo;
    if (j->pending > 0) return;
    if (!j->done) { pthread_cond_broadcast(&g_pool.done); return; }
    {
        void (*done)(void *) = j->done;
        void *arg = j->arg;
        pthread_mutex_unlock(&g_pool.mu);                /* (the callback may free or re-post the job) */
        done(arg);
        pthread_mutex_lock(&g_pool.mu);
    }
}

/* jobs posted so far: with MZ_POOL_SPIN_US=<n> a worker that has run out of work watches this for n microseconds before
 * it goes to sleep.  Off by default: measured on the 50 000-pair C2 call (a job every ~0.3 ms), 0 / 40 / 100 / 300 us all
 * gave 9.4-9.8 ms -- the wake-up of the sleepers is not what the packing waits for */
static unsigned g_posted;
static int g_spin_us = -1;

static void *pool_worker(void *arg)
{
    (void)arg;
    pthread_mutex_lock(&g_pool.mu);
    while (!g_pool.quit) {
        pjob *j;
        int lo, hi;
        if (!(j = grab_any(&lo, &hi))) {
            if (g_spin_us > 0) {
                const unsigned seen = __atomic_load_n(&g_posted, __ATOMIC_RELAXED);
                struct timespec t0, t1;
                pthread_mutex_unlock(&g_pool.mu);
                clock_gettime(CLOCK_MONOTONIC, &t0);
                for (;;) {
                    int k;
                    for (k = 0; k < 64; ++k) __builtin_ia32_pause();
                    if (__atomic_load_n(&g_posted, __ATOMIC_ACQUIRE) != seen) break;
                    clock_gettime(CLOCK_MONOTONIC, &t1);
                    if ((t1.tv_sec - t0.tv_sec) * 1000000L + (t1.tv_nsec - t0.tv_nsec) / 1000L >= g_spin_us) break;
                }
                pthread_mutex_lock(&g_pool.mu);
                if (g_pool.quit) break;
                if ((j = grab_any(&lo, &hi))) goto work;
            }
            pthread_cond_wait(&g_pool.work, &g_pool.mu);
            continue;
        }
work:
        pthread_mutex_unlock(&g_pool.mu);
        j->fn(j->ctx, lo, hi);
        pthread_mutex_lock(&g_pool.mu);
        piece_done(j, lo, hi);
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
    if (want > MZ_COPY_THREADS && !(e && atoi(e) > 0)) want = MZ_COPY_THREADS;
    if (want < 1) want = 1;
    if (want > POOL_MAX) want = POOL_MAX;
    if (g_spin_us < 0) { const char *sp = getenv("MZ_POOL_SPIN_US"); g_spin_us = sp ? atoi(sp) : 0; }
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
    job->next = 0; job->pending = job->n; job->link = NULL;
    if (!g_pool.started) pool_start_locked();
    for (pp = &g_pool.head; *pp; pp = &(*pp)->link) ;
    *pp = job;
    __atomic_fetch_add(&g_posted, 1u, __ATOMIC_RELEASE);
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
        if (hi >= n) { pjob **pp; for (pp = &g_pool.head; *pp != &job; pp = &(*pp)->link) ; *pp = job.link; job.link = NULL; }
        pthread_mutex_unlock(&g_pool.mu);
        fn(ctx, lo, hi);
        pthread_mutex_lock(&g_pool.mu);
        job.pending -= hi - lo;
    }
    while (job.pending > 0) pthread_cond_wait(&g_pool.done, &g_pool.mu);
    pthread_mutex_unlock(&g_pool.mu);
}

/* The same loop, not waited for: returns at once; job->done(job->arg) is called -- by whichever thread finishes the last
 * piece -- when all of [0, n) has been run.  The job must stay where it is until then.  n == 0: done() is called here. */
void mzi_post(mz_ajob *job)
{
    if (job->grain < 1) job->grain = 1;
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
    while (!ready(arg)) {
        pjob *j;
        int lo, hi;
        if (!(j = grab_any(&lo, &hi))) { pthread_cond_wait(&g_pool.work, &g_pool.mu); continue; }
        pthread_mutex_unlock(&g_pool.mu);
        j->fn(j->ctx, lo, hi);
        pthread_mutex_lock(&g_pool.mu);
        piece_done(j, lo, hi);
    }
    pthread_mutex_unlock(&g_pool.mu);
}
void mzi_pool_kick(void)
{
    pthread_mutex_lock(&g_pool.mu);
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
