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
void mz_assemble_cols(int K, int L, int M, int N, const uint8_t *A, const uint8_t *B, const uint8_t *script, int om, uint8_t *out);
void mz_pack_classes_stream(const uint8_t *src, size_t n, uint8_t *dst, size_t slot);
uint32_t mz_pack_band_nib_stream(const int *LB, const int *RB, int M, uint8_t *dst, size_t slot);
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
    /* the host half of mz_yama_batch()'s link formats (mz_pack.c) on exactly-sized heap blocks: the 16-byte copies of
     * the column assembly and the streaming forms must stay inside their arrays (ASan sees every byte beyond) */
    {
        int kk, ll;
        for (kk = 1; kk <= 40; kk += (kk < 6 ? 1 : 17)) for (ll = 1; ll <= 33; ll += (ll < 5 ? 1 : 14)) {
            const int M_ = 37 + kk, N_ = 29 + ll, om = M_ + N_ - 20;      /* 20 aligned columns, the rest one-sided */
            uint8_t *a = malloc((size_t)kk * M_), *b = malloc((size_t)ll * N_), *scr = calloc((size_t)(om + 3) / 4, 1);
            uint8_t *out = malloc((size_t)om * (kk + ll)), *nib;
            int *lb = malloc(sizeof(int) * (M_ + 1)), *rb = malloc(sizeof(int) * (M_ + 1)), m, ia = 0, ib = 0;
            size_t slot;
            memset(a, 'A', (size_t)kk * M_); memset(b, 'c', (size_t)ll * N_);
            for (m = 0; m < om; ++m) {                   /* C while both last, then D's, then I's; 2 bits per column */
                const unsigned op = (m < 20) ? 0u : (ia < M_ ? 2u : 1u);
                scr[m >> 2] |= (uint8_t)(op << (2 * (m & 3)));
                ia += op != 1u; ib += op != 2u;
            }
            if (ia != M_ || ib != N_) { fprintf(stderr, "bad test script\n"); return 1; }
            mz_assemble_cols(kk, ll, M_, N_, a, b, scr, om, out);
            sum += out[0] + out[(size_t)om * (kk + ll) - 1];
            slot = (((size_t)kk * M_ + 63) & ~(size_t)63) / 2;
            nib = aligned_alloc(32, slot);
            mz_pack_classes_stream(a, (size_t)kk * M_, nib, slot);
            sum += nib[0] + nib[slot - 1];
            free(nib);
            for (m = 0; m <= M_; ++m) { lb[m] = m / 2; rb[m] = m / 2 + 11 + (m & 3); }
            for (m = 1; m <= M_; ++m) if (rb[m] < rb[m - 1]) rb[m] = rb[m - 1];
            slot = ((size_t)M_ + 31) & ~(size_t)31;
            nib = aligned_alloc(32, slot);
            sum += mz_pack_band_nib_stream(lb, rb, M_, nib, slot) + nib[slot - 1];
            free(nib);
            free(a); free(b); free(scr); free(out); free(lb); free(rb);
        }
    }
    init_scores70(); init_scores85(); init_scores70();
    printf("san host ok %lld\n", sum);
    return 0;
}
