/* tests/tools/spread/spread.c -- how many bits do the states of ONE band row span?  (would two 16-bit cells per
 * VALU lane -- v_pk_add_i16 / v_pk_max_i16 -- hold the row-parallel kernel's states?)  Runs the profile oracle (a private copy with a
 * counting hook) over synthetic pairs of a shape and prints, as histograms over bit lengths: a row's spread (max - min over its reachable
 * C / D / I states: what re-basing per ROW would leave), a state's step to its neighbour in the row (what difference encoding would hold),
 * a cell's spread (max - min of its own three states).  The kernels keep 4 * value + tag: two more bits than printed.
 *     gcc -O2 -I oracle -I tests/tools/spread -DMZO_SPREAD_STATS tests/tools/spread/spread.c oracle/yama_profile_oracle.c oracle/yama_oracle.c -o scratch/spread
 *     scratch/spread K L M R pairs                                                                                                          */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"
long long sp_rows, sp_cells, sp_hist_row[40], sp_hist_nb[40], sp_hist_cell[40];
int32_t sp_row_lo = 0x7fffffff, sp_row_hi = -0x7fffffff, sp_prev[3]; int sp_have_prev;
static unsigned long long x64 = 88172645463325252ULL;
static unsigned rnd(void) { x64 ^= x64 << 13; x64 ^= x64 >> 7; x64 ^= x64 << 17; return (unsigned)(x64 >> 11); }
static void column(uint8_t *col, int rows) { int i, live = 0; for (i = 0; i < rows; ++i) { col[i] = rnd() % 100 < 8 ? '-' : "ACGT"[rnd() & 3]; live += col[i] != '-'; } if (!live) col[0] = 'A'; }
static void hist(const char *what, const long long *h, long long n)
{
    int b; long long acc = 0;
    printf("%s (n = %lld):", what, n);
    for (b = 0; b < 40; ++b) if (h[b]) { acc += h[b]; printf("  %d bits %.4f%%", b, 100.0 * h[b] / n); }
    printf("\n");
}
int main(int argc, char **argv)
{
    const int K = argc > 1 ? atoi(argv[1]) : 2, L = argc > 2 ? atoi(argv[2]) : 2, M0 = argc > 3 ? atoi(argv[3]) : 1000, R = argc > 4 ? atoi(argv[4]) : 30, pairs = argc > 5 ? atoi(argv[5]) : 50;
    mzo_scores sc;
    int p;
    mzo_scores_hoxd70(&sc);
    for (p = 0; p < pairs; ++p) {
        const int M = M0 * 9 / 10 + (int)(rnd() % (unsigned)(M0 / 5 + 1)), N = M0 * 9 / 10 + (int)(rnd() % (unsigned)(M0 / 5 + 1));
        uint8_t *A = malloc((size_t)K * M), *B = malloc((size_t)L * N), *out = malloc((size_t)(K + L) * (M + N) + 16);
        int *LB = malloc(sizeof(int) * (M + 1)), *RB = malloc(sizeof(int) * (M + 1)), i, j, OM = 0; int32_t fin[3];
        for (i = 0; i < M; ++i) column(A + (size_t)i * K, K);
        for (j = 0; j < N; ++j) {                               /* B: a noisy copy of A, random behind M */
            uint8_t *col = B + (size_t)j * L; int live = 0, k;
            if (j < M) for (k = 0; k < L; ++k) { uint8_t ch = A[(size_t)j * K + k % K]; if (rnd() % 100 < 10) ch = "ACGT"[rnd() & 3]; if (rnd() % 100 < 8) ch = '-'; col[k] = ch; live += ch != '-'; }
            else { column(col, L); live = 1; }
            if (!live) col[0] = 'C';
        }
        for (i = 0; i <= M; ++i) LB[i] = RB[i] = (int)((long long)i * N / M);
        LB[0] = 0;
        mzo_smooth(LB, RB, M, N, R);
        if (mzo_yama_profile(A, K, M, B, L, N, LB, RB, &sc, out, &OM, fin, NULL) != 0) { fprintf(stderr, "pair %d refused\n", p); return 1; }
        free(A); free(B); free(out); free(LB); free(RB);
    }
    printf("K=%d L=%d M,N~%d R=%d, %d pairs: one step moves a state by at most K*L*(400+125) = %d\n", K, L, M0, R, pairs, K * L * 525);
    hist("a band row's spread, max - min of its reachable states", sp_hist_row, sp_rows);
    { long long t = 0; int b; for (b = 0; b < 40; ++b) t += sp_hist_nb[b]; hist("|state(r,c) - state(r,c-1)|, same state", sp_hist_nb, t); }
    hist("a cell's spread, max - min of its reachable C / D / I", sp_hist_cell, sp_cells);
    return 0;
}
