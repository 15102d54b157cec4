/* Host-only entry points of libmzamd.so under ASan/UBSan (tests/test_sanitizers.py; `make -C multiz_amd/csrc san`):
 * band construction, dash-column removal, column mapping, the synthetic generators, segment gathering, scoring. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "../../include/mz_amd.h"
#include "../../include/mz_preyama.h"
#include "../../include/mz_scores.h"

void mz_synth_shapes(int n, uint64_t seed, int64_t first_pair, int K, int L, int mlo, int mhi, int32_t *aK, int32_t *aL, int32_t *aM,
                     int32_t *aN, int64_t *offA, int64_t *offB, int64_t *offBand, int64_t totals[3]);
void mz_synth_shapes_tree(int n, uint64_t seed, int64_t first_pair, int mlo, int mhi, int32_t *aK, int32_t *aL, int32_t *aM,
                          int32_t *aN, int64_t *offA, int64_t *offB, int64_t *offBand, int64_t totals[3]);
void mz_synth_fill(int n, uint64_t seed, int64_t first_pair, int radius, const int32_t *aK, const int32_t *aL, const int32_t *aM,
                   const int32_t *aN, const int64_t *offA, const int64_t *offB, const int64_t *offBand, uint8_t *poolA, uint8_t *poolB,
                   int32_t *poolLB, int32_t *poolRB);
int mz_gather_segments(int64_t n, int64_t elem, const int64_t *off, const int64_t *len, const int64_t *pos, const void *src, void *dst);

int main(void)
{
    enum { n = 300 };
    int32_t K[n], L[n], M[n], N[n];
    int64_t oa[n], ob[n], od[n], tot[3], off[n], len[n], pos[n];
    uint8_t *A, *B, *G;
    int32_t *LB, *RB;
    long long sum = 0;
    int p, i, t;
    for (t = 0; t < 2; ++t) {
        if (t) mz_synth_shapes_tree(n, 7, 100, 20, 300, K, L, M, N, oa, ob, od, tot);
        else mz_synth_shapes(n, 7, 0, 3, 2, 1, 400, K, L, M, N, oa, ob, od, tot);
        A = malloc((size_t)tot[0]); B = malloc((size_t)tot[1]); LB = malloc(4 * (size_t)tot[2]); RB = malloc(4 * (size_t)tot[2]);
        mz_synth_fill(n, 7, t ? 100 : 0, 30, K, L, M, N, oa, ob, od, A, B, LB, RB);
        for (p = 0; p < n; ++p) { off[p] = oa[n - 1 - p]; len[p] = (int64_t)K[n - 1 - p] * M[n - 1 - p]; sum += LB[od[p] + M[p]] + RB[od[p]]; }
        G = malloc((size_t)tot[0]);
        { int64_t acc = 0; for (p = 0; p < n; ++p) { pos[p] = acc; acc += len[p]; } }
        if (mz_gather_segments(n, 1, off, len, pos, A, G) != 0) { fprintf(stderr, "gather refused\n"); return 1; }
        if (memcmp(G, A + oa[n - 1], (size_t)len[0]) != 0) { fprintf(stderr, "gather mismatch\n"); return 1; }
        /* rmColDash + mapping on a pair's columns (1-based pointer arrays over a private copy, one spare byte) */
        for (p = 0; p < n; p += 37) {
            int cols = M[p], kept = cols, *map, *map2;
            unsigned char **X = (unsigned char **)malloc((size_t)cols * sizeof *X) - 1, *buf = malloc((size_t)cols * K[p] + 1);
            memcpy(buf, A + oa[p], (size_t)cols * K[p]); buf[(size_t)cols * K[p]] = 'N';
            for (i = 1; i <= cols; ++i) X[i] = buf + (size_t)(i - 1) * K[p];
            for (i = 3; i <= cols; i += 5) memset(X[i], '-', (size_t)K[p]);
            map = rmColDash(X, &kept, K[p]);
            map2 = mapping(X, 0, K[p] - 1, 1, kept, X, 0, K[p] - 1, 1, kept);
            sum += kept + map[cols] + map2[kept];
            free(map); free(map2); free(buf); free(X + 1);
        }
        free(A); free(B); free(LB); free(RB); free(G);
    }
    init_scores70(); init_scores85(); init_scores70();
    printf("san host ok %lld\n", sum);
    return 0;
}
