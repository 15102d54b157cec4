/* tests/tools/pool_stress.c -- the host thread pool of the chunk pipelines (multiz_amd/csrc/mz_pool.c) alone, no GPU: loops that are
 * waited for (mzi_parallel_for) from several threads at once, loops that are posted (mzi_post) with their completion callbacks, and a
 * caller that works in the pool until a condition holds (mzi_help_until / mzi_pool_kick).  Built with -fsanitize=thread by
 * tests/test_sanitizers.py; prints "pool ok" when every piece of every loop ran exactly once. */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <time.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../multiz_amd/csrc/mz_ctx.h"

#define NJOB 64
#define NITEM 3000
static unsigned char hit[NJOB][NITEM];
static int done_flag[NJOB];
static int n_done;

static void body(void *ctx, int lo, int hi)
{
    unsigned char *h = (unsigned char *)ctx;
    int i;
    volatile unsigned x = 0;
    for (i = lo; i < hi; ++i) { h[i]++; for (int k = 0; k < 50; ++k) x += (unsigned)k; }
}
/* a loop whose pieces may be run twice (job.hedge): every piece stores the same values; piece 3 of it dawdles for 40 ms the first time */
static unsigned char hhit[NITEM];
static int dawdled, hedge_done;
static void hbody(void *ctx, int lo, int hi)
{
    int i;
    (void)ctx;
    if (lo == 3 * 25 && !__atomic_exchange_n(&dawdled, 1, __ATOMIC_ACQ_REL)) { struct timespec t = { 0, 40000000 }; nanosleep(&t, NULL); }
    for (i = lo; i < hi; ++i) __atomic_store_n(&hhit[i], 1, __ATOMIC_RELAXED);
}
static void on_hdone(void *arg) { (void)arg; __atomic_store_n(&hedge_done, 1, __ATOMIC_RELEASE); mzi_pool_kick(); }
static int hedge_ready(void *arg) { (void)arg; return __atomic_load_n(&hedge_done, __ATOMIC_ACQUIRE); }

static void on_done(void *arg)
{
    int j = (int)(long)arg;
    __atomic_store_n(&done_flag[j], 1, __ATOMIC_RELEASE);
    __atomic_fetch_add(&n_done, 1, __ATOMIC_ACQ_REL);
    mzi_pool_kick();
}
static int all_done(void *arg) { return __atomic_load_n(&n_done, __ATOMIC_ACQUIRE) == *(int *)arg; }

static void *sync_caller(void *arg)
{
    unsigned char *h = (unsigned char *)calloc(NITEM, 1);
    int r, i, bad = 0;
    (void)arg;
    for (r = 0; r < 40; ++r) mzi_parallel_for(NITEM, 37, body, h);
    for (i = 0; i < NITEM; ++i) bad += h[i] != 40;
    free(h);
    return (void *)(long)bad;
}

int main(void)
{
    static mz_ajob job[NJOB];
    pthread_t th[3];
    int j, i, bad = 0, want = NJOB, round;
    for (round = 0; round < 3; ++round) {
        memset(hit, 0, sizeof hit); memset(done_flag, 0, sizeof done_flag); n_done = 0;
        for (i = 0; i < 3; ++i) pthread_create(&th[i], NULL, sync_caller, NULL);
        for (j = 0; j < NJOB; ++j) {
            memset(&job[j], 0, sizeof job[j]);
            job[j].fn = body; job[j].ctx = hit[j]; job[j].n = j % 7 == 3 ? 0 : NITEM - j; job[j].grain = 1 + j % 50;
            job[j].done = on_done; job[j].arg = (void *)(long)j;
            mzi_post(&job[j]);
        }
        mzi_help_until(all_done, &want);
        for (i = 0; i < 3; ++i) { void *r; pthread_join(th[i], &r); bad += (int)(long)r; }
        for (j = 0; j < NJOB; ++j) {
            const int n = j % 7 == 3 ? 0 : NITEM - j;
            if (!done_flag[j]) ++bad;
            for (i = 0; i < NITEM; ++i) bad += hit[j][i] != (i < n);
        }
    }
    {   /* the hedged loop: complete well before the dawdler is back (400 us + a piece; the bound here is lenient: a loaded test machine), quiet only after it */
        static mz_ajob hj;
        struct timespec t0, t1, t2;
        int twice;
        memset(&hj, 0, sizeof hj);
        hj.fn = hbody; hj.n = NITEM; hj.grain = 25; hj.done = on_hdone; hj.hedge = 1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        mzi_post(&hj);
        while (!hedge_ready(NULL)) { struct timespec nap = { 0, 50000 }; nanosleep(&nap, NULL); }      /* (not working in the pool: this thread must not be the dawdler) */
        clock_gettime(CLOCK_MONOTONIC, &t1);
        twice = mzi_job_quiet(&hj);
        clock_gettime(CLOCK_MONOTONIC, &t2);
        for (i = 0; i < NITEM; ++i) bad += hhit[i] != 1;
        {
            const double done_ms = 1e3 * (t1.tv_sec - t0.tv_sec) + 1e-6 * (t1.tv_nsec - t0.tv_nsec), quiet_ms = 1e3 * (t2.tv_sec - t0.tv_sec) + 1e-6 * (t2.tv_nsec - t0.tv_nsec);
            printf("hedged loop: complete after %.2f ms, quiet after %.2f ms, %d piece(s) run twice\n", done_ms, quiet_ms, twice);
            if (getenv("MZ_HEDGE_US") == NULL && (twice < 1 || done_ms > 20.0 || quiet_ms < 30.0)) { printf("hedging did not work\n"); ++bad; }
        }
    }
    mzi_pool_stop();
    if (bad) { printf("pool FAILED: %d\n", bad); return 1; }
    printf("pool ok\n");
    return 0;
}
