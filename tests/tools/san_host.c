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
    /* the text path's host assembly (mz_assemble_rows): rows spread by one bit per merged column, squeezed by bits or by their own
     * dashes, on exactly-sized heap blocks at every phase of a 64-column word */
    {
        typedef struct { const uint8_t *src; int n, squeeze; const uint64_t *keep, *ops; uint8_t *tmp; } rowspec;
        void mz_assemble_rows(int nrows, const rowspec *rows, int om, uint8_t *out);
        int om, r;
        for (om = 1; om <= 300; om += (om < 70 ? 1 : 37)) {
            enum { NR = 3 };
            const int words = (om + 63) / 64, nsrc = om + 9;
            uint64_t *ops = calloc((size_t)words, 8), *keep = calloc((size_t)(nsrc + 63) / 64, 8);
            uint8_t *src[NR], *tmp[NR], *out = malloc((size_t)NR * om);
            rowspec rows[NR];
            int taken = 0, kept = 0, live = 0, c;
            for (c = 0; c < om; ++c) if (c % 3 != 1) { ops[c >> 6] |= 1ULL << (c & 63); ++taken; }
            for (r = 0; r < NR; ++r) { src[r] = malloc((size_t)nsrc); tmp[r] = malloc((size_t)nsrc + 16); }
            /* row 0: its first `taken` bytes as they are; row 1: the bytes whose keep bit is set (exactly `taken` of them); row 2: its non-dash bytes */
            for (c = 0; c < nsrc; ++c) {
                src[0][c] = (uint8_t)('A' + c % 4);
                src[1][c] = (uint8_t)('a' + c % 4);
                if (kept < taken && c % 7 != 3) { keep[c >> 6] |= 1ULL << (c & 63); ++kept; }
                src[2][c] = (live < taken && c % 5 != 2) ? (++live, (uint8_t)'G') : (uint8_t)'-';
            }
            if (kept != taken || live != taken) { fprintf(stderr, "bad row test\n"); return 1; }
            rows[0].src = src[0]; rows[0].n = taken; rows[0].squeeze = 0; rows[0].keep = NULL; rows[0].ops = ops; rows[0].tmp = NULL;
            rows[1].src = src[1]; rows[1].n = nsrc; rows[1].squeeze = 1; rows[1].keep = keep; rows[1].ops = ops; rows[1].tmp = tmp[1];
            rows[2].src = src[2]; rows[2].n = nsrc; rows[2].squeeze = 2; rows[2].keep = NULL; rows[2].ops = ops; rows[2].tmp = tmp[2];
            mz_assemble_rows(NR, rows, om, out);
            for (c = 0; c < om; ++c) {
                const int dash = c % 3 == 1;
                if ((out[c] == '-') != dash || (out[om + c] == '-') != dash || (out[2 * om + c] == '-') != dash) { fprintf(stderr, "row assembly wrong at om %d column %d\n", om, c); return 1; }
            }
            sum += out[0] + out[(size_t)NR * om - 1];
            for (r = 0; r < NR; ++r) { free(src[r]); free(tmp[r]); }
            free(ops); free(keep); free(out);
        }
    }
    /* link images (mz_link_pack / mz_link_assemble: host only): jobs -> image, a hand-made result image -> merged columns; an image that
     * is too short, points outside itself, or holds a script that does not take the job's columns is refused */
    {
        enum { NJ = 7 };
        typedef struct { int32_t status, badrow, om, f[3]; int64_t off, cells; } res_rec;
        mz_job jobs[NJ];
        mz_out outs[NJ];
        mz_link_desc d;
        void *img = NULL, *exc = NULL;
        uint8_t *res;
        res_rec *rec;
        size_t at = 0, scripts_at = 64 + ((sizeof(res_rec) * NJ + 255) & ~(size_t)255), bytes;
        int j, m;
        for (j = 0; j < NJ; ++j) {
            const int K_ = 1 + j % 4, L_ = 1 + (j * 3) % 5, M_ = 30 + 9 * j, N_ = 41 + 5 * j;
            uint8_t *a = malloc((size_t)K_ * M_), *b = malloc((size_t)L_ * N_);
            int *lb = malloc(sizeof(int) * (M_ + 1)), *rb = malloc(sizeof(int) * (M_ + 1));
            memset(a, "ACGT-n"[j % 6], (size_t)K_ * M_); memset(b, 't', (size_t)L_ * N_);
            for (m = 0; m <= M_; ++m) { lb[m] = m > 12 ? m - 12 : 0; rb[m] = m + 12 + (j == 3 && m > 9 ? 40 : 0) + (j == 5 && m > 4 ? 700 : 0); if (rb[m] > N_ || m == M_) rb[m] = N_; }
            jobs[j].K = K_; jobs[j].L = L_; jobs[j].M = M_; jobs[j].N = N_; jobs[j].A = a; jobs[j].B = b; jobs[j].LB = lb; jobs[j].RB = rb;
        }
        if (mz_link_pack(NJ, jobs, &d, &img, &exc) != 0 || d.n != NJ || !img) { fprintf(stderr, "mz_link_pack: %s\n", mz_last_error()); return 1; }
        sum += ((uint8_t *)img)[0] + ((uint8_t *)img)[d.image_bytes - 1] + d.exc_bytes;
        bytes = scripts_at + 64;
        for (j = 0; j < NJ; ++j) bytes += (((size_t)(jobs[j].M + jobs[j].N) - 20 + 3) / 4 + 3) & ~(size_t)3;
        res = calloc(bytes, 1);
        rec = (res_rec *)(res + 64);
        for (j = 0; j < NJ; ++j) {
            const int om = jobs[j].M + jobs[j].N - 20;
            int ia = 0, ib = 0;
            rec[j].status = j == 2 ? MZ_E_NARROW : MZ_OK; rec[j].badrow = j == 2 ? 5 : -1; rec[j].om = j == 2 ? 0 : om; rec[j].off = (int64_t)at; rec[j].cells = 100;
            for (m = 0; m < om; ++m) {
                const unsigned op = (m < 20) ? 0u : (ia < jobs[j].M ? 2u : 1u);
                res[scripts_at + at + (m >> 2)] |= (uint8_t)(op << (2 * (m & 3)));
                ia += op != 1u; ib += op != 2u;
            }
            at += (((size_t)om + 3) / 4 + 3) & ~(size_t)3;
        }
        if (mz_link_assemble(NJ, jobs, res, (int64_t)bytes, outs) != 1) { fprintf(stderr, "mz_link_assemble: %s\n", mz_last_error()); return 1; }
        for (j = 0; j < NJ; ++j) if (j != 2) sum += outs[j].cols[0] + outs[j].cols[(size_t)outs[j].OM * (jobs[j].K + jobs[j].L) - 1];
        if (outs[2].status != MZ_E_NARROW || outs[2].cols) { fprintf(stderr, "a refused pair got columns\n"); return 1; }
        mz_free_outs(NJ, outs);
        if (mz_link_assemble(NJ, jobs, res, (int64_t)scripts_at - 8, outs) != -1) { fprintf(stderr, "a short image was accepted\n"); return 1; }
        {   /* a record of the right size whose script takes a column of A at EVERY step: more columns of A than the job has (the
             * assembly would read past the caller's A) */
            const int om = jobs[4].M + jobs[4].N - 20;
            uint8_t *keep = malloc(((size_t)om + 3) / 4);
            memcpy(keep, res + scripts_at + rec[4].off, ((size_t)om + 3) / 4);
            memset(res + scripts_at + rec[4].off, 0xAA, ((size_t)om + 3) / 4);          /* D D D D ... */
            if (mz_link_assemble(NJ, jobs, res, (int64_t)bytes, outs) != -1) { fprintf(stderr, "a script that does not fit its job was accepted\n"); return 1; }
            memcpy(res + scripts_at + rec[4].off, keep, ((size_t)om + 3) / 4);
            free(keep);
            if (mz_link_assemble(NJ, jobs, res, (int64_t)bytes, outs) != 1) { fprintf(stderr, "the restored image was refused: %s\n", mz_last_error()); return 1; }
            mz_free_outs(NJ, outs);
        }
        rec[4].off = (int64_t)bytes;
        if (mz_link_assemble(NJ, jobs, res, (int64_t)bytes, outs) != -1) { fprintf(stderr, "an image that points outside itself was accepted\n"); return 1; }
        free(res); mz_link_free(img); mz_link_free(exc);
        for (j = 0; j < NJ; ++j) { free((void *)jobs[j].A); free((void *)jobs[j].B); free((void *)jobs[j].LB); free((void *)jobs[j].RB); }
    }
    init_scores70(); init_scores85(); init_scores70();
    printf("san host ok %lld\n", sum);
    return 0;
}
