/* oracle/ref_batch.c -- OpenMP batch driver around the COMPILED REFERENCE (oracle/_ref/libref.so).
 *
 * TEST INFRASTRUCTURE ONLY.  dlopen()s the reference library built by oracle/Makefile from
 * /root/reference (no reference source is copied here), calls its own init_scores70() and
 * yama() (reference mz_yama.h:22) on every pair of a packed batch -- one pair per thread; yama()
 * only reads the score globals and allocates its own scratch, so concurrent calls are safe -- and
 * hashes (OM, merged columns) exactly like mzo_yama_batch().  Used for bench.py's cpu_baseline
 * leg with kind = "reference" and for cross-checking the restatement at benchmark sizes.
 */
#include <dlfcn.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "oracle.h"

#pragma GCC diagnostic ignored "-Wfree-nonheap-object"   /* 1-based arrays: free(X + 1) */

typedef void (*yama_fn)(unsigned char **A, int K, int M, unsigned char **B, int L, int N,
                        int *LB, int *RB, unsigned char ***OAL, int *OM);

/* returns number of pairs rejected by the validity prologue (the reference would exit(1) on
 * those, so they are filtered here), or -1 if the library cannot be loaded */
int mzo_ref_batch(const char *libref_path, int n, const int *K, const int *L, const int *M, const int *N,
                  const int64_t *offA, const int64_t *offB, const int64_t *offBand,
                  const uint8_t *poolA, const uint8_t *poolB, const int *poolLB, const int *poolRB,
                  int threads, int *om, uint64_t *hash, int64_t *cells_done)
{
    void *h = dlopen(libref_path, RTLD_NOW | RTLD_LOCAL);
    void (*init70)(void);
    yama_fn ref_yama;
    int bad = 0, p;
    int64_t total = 0;

    if (!h) return -1;
    init70 = (void (*)(void))dlsym(h, "init_scores70");
    ref_yama = (yama_fn)dlsym(h, "yama");
    if (!init70 || !ref_yama) return -1;
    init70();
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel for schedule(dynamic, 1) reduction(+:bad, total)
    for (p = 0; p < n; ++p) {
        const int m = M[p], nn = N[p], k = K[p], l = L[p];
        int64_t cells = 0;
        if (mzo_yama_check(m, nn, poolLB + offBand[p], poolRB + offBand[p], &cells, NULL)) {
            bad++; om[p] = -1; hash[p] = 0;
            continue;
        }
        {
            /* the reference's calling convention: 1-based arrays of column pointers */
            unsigned char **A = (unsigned char **)malloc(sizeof(void *) * (size_t)m) - 1;
            unsigned char **B = (unsigned char **)malloc(sizeof(void *) * (size_t)nn) - 1;
            unsigned char **out = NULL;
            int i, m_new = 0;
            uint64_t hh;
            for (i = 1; i <= m; ++i)  A[i] = (unsigned char *)(poolA + offA[p] + (size_t)(i - 1) * k);
            for (i = 1; i <= nn; ++i) B[i] = (unsigned char *)(poolB + offB[p] + (size_t)(i - 1) * l);
            ref_yama(A, k, m, B, l, nn, (int *)(poolLB + offBand[p]), (int *)(poolRB + offBand[p]), &out, &m_new);
            hh = mzo_fnv1a((const uint8_t *)&m_new, 4, 0);
            om[p] = m_new;
            hash[p] = mzo_fnv1a(out[1], (int64_t)m_new * (k + l), hh);
            total += cells;
            {   /* two blocks, freed as the reference documents (mz_yama.h:17-18) */
                void *cols = out[1], *ptrs = (void *)(out + 1);
                free(cols); free(ptrs);
            }
            { void *pa = (void *)(A + 1), *pb = (void *)(B + 1); free(pa); free(pb); }
        }
    }
    if (cells_done) *cells_done = total;
    return bad;
}

/* `threads` threads each run the same register-only integer chain of `iters` steps: wall seconds.  A box that gives
 * every thread a core takes the same time at any thread count; a CPU quota, SMT sharing or oversubscribed vCPUs show
 * up here without any memory traffic (tests/tools/cpu_scaling.py, bench.py's cpu_baseline.scaling). */
#include <time.h>
double mzo_spin(int threads, int64_t iters)
{
    struct timespec t0, t1;
    uint64_t sink = 0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
#pragma omp parallel num_threads(threads) reduction(+:sink)
    {
        uint64_t x = 88172645463325252ULL + (uint64_t)
#ifdef _OPENMP
            omp_get_thread_num();
#else
            0;
#endif
        int64_t i;
        for (i = 0; i < iters; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; }
        sink += x;
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    return (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec) + (sink == 42 ? 1e-12 : 0.0);
}

/* per-pair hash of (OM, merged columns) as mzo_ref_batch() / mzo_yama_batch() compute it, over columns that lie
 * anywhere in memory (the mz_out.cols pointers of the product's host path): bench.py's parity gate */
void mzo_hash_cols(int n, const uint64_t *cols_ptr, const int32_t *om, const int32_t *width, uint64_t *hash)
{
    int p;
#pragma omp parallel for schedule(static)
    for (p = 0; p < n; ++p) {
        const int32_t m = om[p];
        const uint64_t h = mzo_fnv1a((const uint8_t *)&m, 4, 0);
        hash[p] = cols_ptr[p] ? mzo_fnv1a((const uint8_t *)(uintptr_t)cols_ptr[p], (int64_t)m * width[p], h) : 0;
    }
}
