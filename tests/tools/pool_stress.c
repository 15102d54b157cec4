/* tests/tools/pool_stress.c -- the host thread pool of the chunk pipelines (multiz_amd/csrc/mz_pool.c) alone, no GPU: loops that are
 * waited for (mzi_parallel_for) from several threads at once, loops that are posted (mzi_post) with their completion callbacks, and a
 * caller that works in the pool until a condition holds (mzi_help_until / mzi_pool_kick).  Built with -fsanitize=thread by
 * tests/test_sanitizers.py; prints "pool ok" when every piece of every loop ran exactly once. */
#include <pthread.h>
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
    mzi_pool_stop();
    if (bad) { printf("pool FAILED: %d\n", bad); return 1; }
    printf("pool ok\n");
    return 0;
}
