/* mz_batch.c -- mz_yama_batch(): N independent yama() calls (reference mz_yama.h:22, mz_yama.c:50-320) from host
 * buffers to host buffers.
 *
 * What bounds this path is the PCIe link (57 GB/s in sum of both directions on this platform) and the host's own
 * memory traffic, not the kernels, so the link carries only what the device needs (mz_pack.c): byte classes of the
 * columns, band bounds as steps, and -- back -- a 32-byte record and a 2-bit edit script per pair.  The merged columns
 * (reference mz_yama.c:293-313) are assembled on the host from the caller's own A and B.
 *
 * A call is cut into chunks that go through a three-stage pipeline on MZ_SETS rotating sets of buffers and streams:
 *
 *   packer     (the calling thread)  classes + band steps into pinned memory on all host threads; issues the copy to the
 *                                    device, the expansion kernels, the plan and the copy of the plan's totals;
 *   launcher   (helper thread)       waits for the totals, sizes the workspaces, issues DP / walk / script packing and
 *                                    the copy of the results;
 *   collector  (helper thread)       waits for the results, allocates ONE block for the chunk's merged columns and
 *                                    assembles them on all host threads.
 *
 * No stage waits for a copy or a kernel another stage could work beside; the chunks' kernels overlap on the GPU
 * through their streams.  A call of one chunk (the drop-in yama(): a batch of one) runs the three steps inline.
 */
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <xmmintrin.h>

#include "mz_ctx.h"
#include "mz_pack.h"

struct asm_ctx;

/* ------------------------------------------------------------------------------------------------ one chunk */

typedef struct pack_ctx {
    const mz_job *jobs;
    const int64_t *hoA, *hoB;
    int64_t *hoC;
    uint8_t *hA, *hB, *hC, *hFmt, *hE;
    int32_t *hLB0, *hRB0;                  /* (NULL: the caller has filled them in) */
    uint32_t *esz;
    /* Where a pair's band steps go in hC, in memory nobody writes after the chunk is cut (NULL: hoC).  chunk_send() turns hoC[p] -- which
     * lies in the staging block and goes to the device -- into the pair's offset in the EXCEPTION block for the pairs that have one, and a
     * piece of the packing that is run a second time (mz_pool.c) may still be at work then: it took that offset for the hC slot, streamed
     * the pair's steps over somebody else's -- or, the offset being no multiple of 16, faulted (one `bench.py --config c4i` in twenty). */
    const int64_t *slotC;
} pack_ctx;

typedef struct chunk {
    mz_ctx *X;
    int set, n, index, lane, wide;         /* lane: which stream slot (X->qf / qd / qt[lane]); -1: the context's own stream (a call of one chunk); wide: a call on four DP slots (its tail is the slot's own) */
    const mz_job *jobs;
    mz_out *outs;
    mz_dev_batch b;
    pack_ctx pc;
    /* the staging block's device-side pointers that are not part of b */
    const int32_t *dLen, *dLB0, *dRB0;
    const int64_t *doC;
    const uint8_t *dFmt, *dC, *dA;
    size_t eA, eB;
    /* the assembling (results_prepare / assemble_range) */
    size_t *where;
    struct asm_ctx *ac;
    int64_t in_bytes, exc_bytes, res_bytes, cells;
    double t_cut0, t_cut1, t_packed, t_send1, t_launch0, t_launch1, t_launch2, t_col0, t_col1, t_col2;
} chunk;

/* a job whose arrays the host may read: the packing loops run before the device's validity prologue (a NULL array is a
 * shape error of the call, MZ_E_SHAPE: the plan sees M = 0) */
static int job_ok(const mz_job *j) { return j->K >= 1 && j->L >= 1 && j->M >= 1 && j->N >= 1 && j->K <= 255 && j->L <= 255 && j->A && j->B && j->LB && j->RB; }
/* (K, L > 255: the plan refuses the pair (MZ_E_ROWS) from K and L alone; its columns do not travel) */

/* expanded bytes of a block's columns on the device: whole 64-byte groups, so that the class nibbles of a pair fill
 * whole 32-byte lines of the staging block (streaming stores, mz_pack.c) */
static size_t cols_padded(int rows, int cols) { return ((size_t)rows * (size_t)cols + 63) & ~(size_t)63; }
static size_t band_slot(int M) { return ((size_t)M + 31) & ~(size_t)31; }      /* a byte per row 1..M; LB[0], RB[0] travel in the header */

/* pairs per piece of the pack / assemble loops: about four pieces per host thread, at most 64 pairs (a chunk of 84
 * long pairs -- BASELINE config 5 -- in pieces of 64 kept two threads busy and took 13 ms to assemble) */
static int pack_grain(int n)
{
    if (n <= 16) return n;                               /* (a single yama() call: the caller alone, no pool is started) */
    const int g = n / (4 * mzi_pool_threads());
    return g < 1 ? 1 : g > 64 ? 64 : g;
}

static void pack_range(void *ctx, int lo, int hi)
{
    const pack_ctx *q = (const pack_ctx *)ctx;
    int p;
    for (p = lo; p < hi; ++p) {
        const mz_job *j = &q->jobs[p];
        uint32_t esz = 0;                                    /* (written once, at the end: a piece that is run a second time -- mz_pool.c -- while the */
        uint8_t fmt = 2;                                     /*  chunk is already on its way must not show anybody a value in between) */
        if (p + 1 < hi) {
            /* the next pair's four arrays start on pages of their own, where no hardware stream is running yet: ask for
             * their first lines now (LB and RB of a C2 pair are a 4 KB page each; the demand misses at every array's head
             * were a third of the packing's time) */
            const mz_job *nx = &q->jobs[p + 1];
            if (job_ok(nx)) {
                int k;
                for (k = 0; k < 256; k += 64) {
                    _mm_prefetch((const char *)nx->LB + k, _MM_HINT_T0); _mm_prefetch((const char *)nx->RB + k, _MM_HINT_T0);
                    _mm_prefetch((const char *)nx->A + k, _MM_HINT_T0); _mm_prefetch((const char *)nx->B + k, _MM_HINT_T0);
                }
            }
        }
        if (q->hLB0) { const int ok = job_ok(j); q->hLB0[p] = ok ? j->LB[0] : 0; q->hRB0[p] = ok ? j->RB[0] : 0; }     /* (an invalid job: one dummy entry, LB[0] = RB[0] = 0) */
        if (job_ok(j)) {
            uint32_t steps;
            mz_pack_classes_stream(j->A, (size_t)j->K * j->M, q->hA + q->hoA[p] / 2, cols_padded(j->K, j->M) / 2);
            mz_pack_classes_stream(j->B, (size_t)j->L * j->N, q->hB + q->hoB[p] / 2, cols_padded(j->L, j->N) / 2);
            steps = mz_pack_band_nib_stream(j->LB, j->RB, j->M, q->hC + (q->slotC ? q->slotC[p] : q->hoC[p]), band_slot(j->M));
            if (steps >= 16u) {                              /* two bytes per row, or raw: through the exception block */
                fmt = steps < 256u ? 1 : 0;
                esz = steps < 256u ? (uint32_t)((8 + 2 * (size_t)j->M + 3) & ~(size_t)3) : (uint32_t)(8 * ((size_t)j->M + 1));
            }
        }
        q->esz[p] = esz; q->hFmt[p] = fmt;
    }
    _mm_sfence();                                            /* the streaming stores are out before the piece is reported done */
}

static void pack_exceptions(void *ctx, int lo, int hi)
{
    const pack_ctx *q = (const pack_ctx *)ctx;
    int p;
    for (p = lo; p < hi; ++p) {
        const mz_job *j = &q->jobs[p];
        if (!q->esz[p]) continue;
        if (q->hFmt[p] == 1) mz_pack_band_bytes(j->LB, j->RB, j->M, q->hE + q->hoC[p]);
        else {
            memcpy(q->hE + q->hoC[p], j->LB, 4 * ((size_t)j->M + 1));
            memcpy(q->hE + q->hoC[p] + 4 * ((size_t)j->M + 1), j->RB, 4 * ((size_t)j->M + 1));
        }
    }
}

static int g_timing = -1;                  /* mzi_timing(): 1 = one JSON line per call, 2 = and one per chunk (stderr) */
#define TSTAMP(X, set, k, st) do { if (g_timing >= 2 && (X)->btime_ready) HIPCK(hipEventRecord((X)->btime[set][k], st)); } while (0)

/* Pieces a GPU's share of a call is cut into (MZ_CHUNKS overrides): chunks of about 5 000 pairs, three to sixteen of them.  A chunk's DP
 * is a launch of its own, and below ~4 000 waves a launch is latency-bound -- it takes as long as its longest pair whatever their
 * number -- while every chunk costs a dozen dependent launches at either end; above ~8 000 the first chunk starts late and the last
 * one ends late.  Measured with the round-5 pipeline (ms per call, parts 3 / 4 / 6 / 8 / 12 / 16): 20 000 pairs with indel bands (c2i)
 * 5.9 / 5.5 / 6.3 / 6.4 / 7.6; 50 000 C2 pairs - / - / - / 8.7 / 8.2; a guide tree's 125 000 (c4) - / - / - / 17.5 / 16.8 / 15.7. */
static int chunk_parts(int n)
{
    static int v = -1;
    int parts;
    if (v < 0) { const char *e = getenv("MZ_CHUNKS"); v = e && atoi(e) > 0 ? atoi(e) : 0; }
    if (v) return v;
    parts = (n + 2500) / 5000;
    parts = parts < 3 ? 3 : parts > 16 ? 16 : parts;
    /* few pairs (long ones as likely as not: the chunks are cut by bytes): their DPs are as long as one pair takes however few they hold, and
     * run two abreast -- an even number of them (1 000 pairs of 100 000 columns, 2 / 3 / 4 chunks: 50.8 / 54.6 / 50.3 ms) */
    if (n <= 4096 && (parts & 1)) ++parts;
    return parts;
}

/* the chunk's streams (mz_ctx.h): front (staging block -> device, expansion, plan), DP, tail (walk, script packing, results -> host) */
static hipStream_t chunk_front(const chunk *c) { return c->lane < 0 ? c->X->stream : c->X->qf[c->index % c->X->nf]; }
static hipStream_t chunk_dp(const chunk *c) { return c->lane < 0 ? c->X->stream : c->X->qd[c->lane]; }
static hipStream_t chunk_tail(const chunk *c) { return c->lane < 0 ? c->X->stream : c->wide ? c->X->qt[c->lane] : c->X->qt[c->index % c->X->nt]; }

/* The calling thread: chunk `index` = the n jobs at `jobs`, laid out in buffer set `set`; its packing as a loop (*pack).
 * One pinned staging block: [K L M N](int32 x n) [offA offB offBand](int64 x n) [bandLen LB0 RB0](int32 x n) offC(int64 x n)
 * fmt(n), band steps, class nibbles of A, class nibbles of B -- every part at a multiple of 256 bytes. */
static int chunk_cut(mz_ctx *X, chunk *c, int index, int set, int lane, int n, const mz_job *jobs, mz_out *outs, mz_ajob *pack)
{
    mz_dev_batch b;
    size_t eA = 0, eB = 0, nband = 0, bytesC = 0, nprep = 0, hdr, in_bytes;
    char *h, *d;
    int32_t *hK, *hL, *hM, *hN, *hLen, *hLB0, *hRB0;
    int64_t *hoA, *hoB, *hoBand, *hoC;
    uint8_t *hFmt, *hC, *hA, *hB;
    int p;

    c->t_cut0 = mzi_now_s();
    c->X = X; c->set = set; c->index = index; c->lane = lane & 0xff; c->wide = lane >= 0 && ((lane >> 8) & 1); c->n = n; c->jobs = jobs; c->outs = outs;
    if (lane < 0) c->lane = -1;
    c->where = NULL; c->ac = NULL; c->cells = 0; c->exc_bytes = 0; c->res_bytes = 0;
    for (p = 0; p < n; ++p) {
        const mz_job *j = &jobs[p];
        if (job_ok(j)) { eA += cols_padded(j->K, j->M); eB += cols_padded(j->L, j->N); nband += (size_t)j->M + 1; bytesC += band_slot(j->M); nprep += MZ_PREP_BOUND(j->N); }
        else { nband += 1; }
    }
    hdr = mzi_al256(4 * (size_t)n) * 7 + mzi_al256(8 * (size_t)n) * 4 + mzi_al256((size_t)n);
    in_bytes = hdr + mzi_al256(bytesC) + mzi_al256(eA / 2) + mzi_al256(eB / 2);
    free((void *)c->pc.slotC);                               /* (one block: the slots, then the exception sizes) */
    c->pc.slotC = (const int64_t *)malloc((size_t)n * sizeof(int64_t) + ((size_t)n + 1) * sizeof *c->pc.esz);
    if (!c->pc.slotC) { c->pc.esz = NULL; return mzi_set_err("out of memory"); }
    c->pc.esz = (uint32_t *)(c->pc.slotC + n);
    if (mzi_host_reserve(&X->h_in[set], in_bytes) || mzi_dev_reserve(&X->d_in[set], in_bytes) ||
        mzi_dev_reserve(&X->d_cols[set], 2 * (mzi_al256(eA / 2) + mzi_al256(eB / 2)) + 256) ||
        mzi_dev_reserve(&X->d_band[set], 2 * mzi_al256(4 * nband)) ||
        mzi_dev_reserve(&X->d_plan[set], mz_dev_plan_bytes(n)) || mzi_host_reserve(&X->h_tot[set], 16 * sizeof(int64_t)) ||
        mzi_dev_reserve(&X->d_prep[set], 4 * nprep + 256)) return -1;
    h = (char *)X->h_in[set].p; d = (char *)X->d_in[set].p;

    memset(&b, 0, sizeof b);
    b.n = n;
    b.dp_hint = lane >= 0 ? MZ_DP_HELPERS_FIRST | ((lane >> 9) & 1 ? MZ_DP_SOLO : 0) : 0;      /* (beside other chunks' DPs: include/mz_amd.h; lane bit 9: a call of at most 1 024 long pairs on four slots) */
#define SLICE(hptr, type, field, bytes) do { hptr = (type *)h; b.field = (const type *)d; \
        h += mzi_al256(bytes); d += mzi_al256(bytes); } while (0)
#define SLICE2(hptr, dptr, type, bytes) do { hptr = (type *)h; dptr = (const type *)d; \
        h += mzi_al256(bytes); d += mzi_al256(bytes); } while (0)
    SLICE(hK, int32_t, K, 4 * (size_t)n); SLICE(hL, int32_t, L, 4 * (size_t)n);
    SLICE(hM, int32_t, M, 4 * (size_t)n); SLICE(hN, int32_t, N, 4 * (size_t)n);
    SLICE(hoA, int64_t, offA, 8 * (size_t)n); SLICE(hoB, int64_t, offB, 8 * (size_t)n);
    SLICE(hoBand, int64_t, offBand, 8 * (size_t)n);
    SLICE2(hLen, c->dLen, int32_t, 4 * (size_t)n);
    SLICE2(hLB0, c->dLB0, int32_t, 4 * (size_t)n); SLICE2(hRB0, c->dRB0, int32_t, 4 * (size_t)n);
    SLICE2(hoC, c->doC, int64_t, 8 * (size_t)n);
    SLICE2(hFmt, c->dFmt, uint8_t, (size_t)n);
    SLICE2(hC, c->dC, uint8_t, bytesC);
    SLICE2(hA, c->dA, uint8_t, eA / 2);
    hB = (uint8_t *)h;
#undef SLICE
#undef SLICE2
    /* the pools the kernels read: A and B expanded to a byte per class, exactly twice the nibble layout */
    b.poolA = (const uint8_t *)X->d_cols[set].p;
    b.poolB = b.poolA + 2 * mzi_al256(eA / 2);
    b.poolLB = (const int32_t *)X->d_band[set].p;
    b.poolRB = (const int32_t *)((char *)X->d_band[set].p + mzi_al256(4 * nband));
    {
        size_t oa = 0, ob = 0, oband = 0, oc = 0;
        for (p = 0; p < n; ++p) {
            const mz_job *j = &jobs[p];
            const int ok = job_ok(j);
            const int nul = !j->A || !j->B || !j->LB || !j->RB;                 /* (reported as MZ_E_SHAPE: the plan sees M = N = 0) */
            hK[p] = j->K; hL[p] = j->L; hM[p] = nul ? 0 : j->M; hN[p] = nul ? 0 : j->N;
            hoA[p] = (int64_t)oa; hoB[p] = (int64_t)ob; hoBand[p] = (int64_t)oband; hoC[p] = (int64_t)oc; ((int64_t *)c->pc.slotC)[p] = (int64_t)oc;
            hLen[p] = ok ? j->M + 1 : 1;                                        /* (LB[0], RB[0]: pack_range -- the first touch of the pair's arrays is the pool's) */
            if (ok) { oa += cols_padded(j->K, j->M); ob += cols_padded(j->L, j->N); oband += (size_t)j->M + 1; oc += band_slot(j->M); }
            else { oband += 1; }
        }
    }
    mz_dev_carve(&b, X->d_plan[set].p);
    b.capTb = b.capScript = b.capOut = INT64_MAX;               /* sizes are not known yet -- except the prep records': at most MZ_PREP_BOUND(N) */
    b.prep = (uint32_t *)X->d_prep[set].p; b.capPrep = (int64_t)(X->d_prep[set].cap / 4);      /* dwords per pair, so they are made with the plan (chunk_send) */
    c->b = b; c->eA = eA; c->eB = eB; c->in_bytes = (int64_t)in_bytes;
    /* the packing, for the pool's threads: byte classes two per byte (mz_pack_classes), band bounds a byte per row where every
     * step is below 16 (the others are noted and sent apart: chunk_send) */
    c->pc.jobs = jobs; c->pc.hoA = hoA; c->pc.hoB = hoB; c->pc.hoC = hoC; c->pc.hA = hA; c->pc.hB = hB; c->pc.hC = hC; c->pc.hFmt = hFmt; c->pc.hE = NULL;
    c->pc.hLB0 = hLB0; c->pc.hRB0 = hRB0;
    pack->fn = pack_range; pack->ctx = &c->pc; pack->n = n; pack->grain = pack_grain(n); pack->hedge = 1;
    c->t_cut1 = mzi_now_s();
    return 0;
}

/* stage 1: the chunk is packed -- the bands that do not fit a nibble per step, then staging block -> device (a kernel on the chunk's
 * stream: mzk_link_copy), expansion, plan, the plan's totals -> host */
static int chunk_send(chunk *c)
{
    mz_ctx *X = c->X;
    const int set = c->set, n = c->n;
    hipStream_t st = chunk_front(c);
    const mz_dev_batch *b = &c->b;
    size_t nexc = 0, bytesE = 0;
    int p;

    c->t_packed = mzi_now_s();
    for (p = 0; p < n; ++p) if (c->pc.esz[p]) { ++nexc; bytesE += c->pc.esz[p]; }
    if (nexc) {
        size_t oe = 0;
        bytesE = (bytesE + 15) & ~(size_t)15;
        if (mzi_host_reserve(&X->h_exc[set], bytesE) || mzi_dev_reserve(&X->d_exc[set], bytesE)) return -1;
        for (p = 0; p < n; ++p) if (c->pc.esz[p]) { c->pc.hoC[p] = (int64_t)oe; oe += c->pc.esz[p]; }
        c->pc.hE = (uint8_t *)X->h_exc[set].p;
        mzi_parallel_for(n, pack_grain(n), pack_exceptions, &c->pc);
    }
    c->exc_bytes = (int64_t)bytesE;
    {
        hipStream_t sc = c->lane < 0 ? st : X->qc;           /* the link copies: a stream of their own (mz_flow.c: which pipe) */
        TSTAMP(X, set, 0, sc);
        if (mzk_link_copy(X->d_in[set].p, X->h_in[set].p, mzi_al256((size_t)c->in_bytes), sc) ||
            (nexc && mzk_link_copy(X->d_exc[set].p, X->h_exc[set].p, bytesE, sc))) return mzi_set_err("%s", mzk_last_error());
        if (sc != st) { HIPCK(hipEventRecord(X->bcopy[set], sc)); HIPCK(hipStreamWaitEvent(st, X->bcopy[set], 0)); }
    }
    if (mzk_unband(n, c->dLen, c->dLB0, c->dRB0, b->offBand, c->doC, c->dFmt, c->dC, (const uint8_t *)X->d_exc[set].p, (int32_t *)b->poolLB, (int32_t *)b->poolRB, st) ||
        mzk_unnib(c->dA, (void *)b->poolA, (long long)(2 * (mzi_al256(c->eA / 2) + mzi_al256(c->eB / 2))), st))
        return mzi_set_err("%s", mzk_last_error());
    TSTAMP(X, set, 1, st);
    /* the totals go home before the prep records are made: the launcher can size the workspaces meanwhile */
    if (mzk_plan(b, st) || mzk_link_copy(X->h_tot[set].p, b->totals, 16 * sizeof(int64_t), st)) return mzi_set_err("%s", mzk_last_error());
    HIPCK(hipEventRecord(X->bplan[set], st));
    if (mzk_prep(b, st)) return mzi_set_err("%s", mzk_last_error());
    HIPCK(hipEventRecord(X->bprep[set], st));
    TSTAMP(X, set, 2, st);
    c->t_send1 = mzi_now_s();
    return 0;
}

/* stage 2: wait for the plan's totals, size the workspaces; the DP on the slot's DP stream (the plan is through -- the host has
 * seen its totals -- so nothing there has to wait for the front stream), walk / script packing / results -> host on the tail stream
 * behind the DP's event */
static int chunk_launch(chunk *c)
{
    mz_ctx *X = c->X;
    const int set = c->set, n = c->n;
    hipStream_t sd = chunk_dp(c), st = chunk_tail(c);
    mz_dev_batch b = c->b;
    const int64_t *totals = (const int64_t *)X->h_tot[set].p;
    size_t res_bytes;
    char *dres;

    c->t_launch0 = mzi_now_s();
    HIPCK(hipEventSynchronize(X->bplan[set]));
    c->t_launch1 = mzi_now_s();

    /* results: 64-byte header, a record per pair, the scripts at a quarter of the plan's script slices */
    res_bytes = mzi_al256(64 + mzi_al256(sizeof(mz_res_rec) * (size_t)n) + (size_t)totals[1] / 4 + 64);
    if (mzi_dev_reserve(&X->d_tb[set], 4 * (size_t)totals[0] + 256) || mzi_dev_reserve(&X->d_script[set], (size_t)totals[1] + 256) ||
        mzi_dev_reserve(&X->d_res[set], res_bytes) || mzi_host_reserve(&X->h_res[set], res_bytes))
        return -1;
    b.tbw = (uint32_t *)X->d_tb[set].p; b.script = (uint8_t *)X->d_script[set].p; b.out = NULL;     /* (no merged columns on the device) */
    b.walk_hint = mz_walk_choice(n, totals);             /* (the plan's totals are here: no need for both launches) */
    b.dp_hint = mz_dp_hint(n, totals) | (b.dp_hint & MZ_DP_REQUESTS);      /* (nor for DP kernels that have no pairs) */
    b.dp_grid = mz_dp_grid(n, totals); b.dp_rows = mz_dp_rows(n, totals); b.hint_gen = g_hint_gen;
    b.capTb = (int64_t)(X->d_tb[set].cap / 4); b.capScript = (int64_t)X->d_script[set].cap; b.capOut = INT64_MAX;
    if (c->lane >= 0 && !X->lanes_made && mz_dp_kinds(b.dp_hint) > 1 && mzi_flow_lanes(X)) return -1;     /* several kinds of pairs: the DP streams' lanes */

    dres = (char *)X->d_res[set].p;
    /* (the prep records: made behind the plan on the front stream -- a few microseconds after the totals this thread has just waited for.
     *  Waited for HERE, not by the DP stream: a wait packet in front of the DP is one more packet that starts 60-200 us late while the
     *  stream's pipe-mate, the tail of the chunk before, has a kernel running; the DP's "done" event rides on its own packet likewise) */
    if (sd != chunk_front(c)) HIPCK(hipEventSynchronize(X->bprep[set]));
    {
        int rides = 0;
        if (mzk_dp_range_ev(&b, 0, n, sd, c->lane < 0 ? NULL : &X->qlane[c->lane], st != sd && g_timing < 2 ? (void *)X->bdp[set] : NULL, &rides)) return mzi_set_err("%s", mzk_last_error());
        TSTAMP(X, set, 3, sd);
        if (st != sd) { if (!rides) HIPCK(hipEventRecord(X->bdp[set], sd)); HIPCK(hipStreamWaitEvent(st, X->bdp[set], 0)); }
    }
    if (mzk_walk(&b, st, 1) || mzk_script_pack(&b, dres, dres + 64, dres + 64 + mzi_al256(sizeof(mz_res_rec) * (size_t)n), st))
        return mzi_set_err("%s", mzk_last_error());
    TSTAMP(X, set, 4, st);
    /* the results go home as the chunk's last kernel: no copy engine, nothing that could wait for anything but this chunk's
     * own kernels */
    if (mzk_link_copy(X->h_res[set].p, dres, res_bytes, st)) return mzi_set_err("%s", mzk_last_error());
    TSTAMP(X, set, 5, st);
    HIPCK(hipEventRecord(X->bdone[set], st));
    c->b = b; c->res_bytes = (int64_t)res_bytes;
    c->t_launch2 = mzi_now_s();
    return 0;
}

typedef struct asm_ctx {
    const mz_job *jobs;
    mz_out *outs;
    const mz_res_rec *rec;
    const uint8_t *packed;
    const size_t *where;
    uint8_t *block;
    int failed;
    int64_t cells;
} asm_ctx;

/* (a piece may be run twice -- mz_pool.c -- so it adds nothing up: results_count() does, afterwards) */
static void assemble_range(void *ctx, int lo, int hi)
{
    asm_ctx *q = (asm_ctx *)ctx;
    int p;
    for (p = lo; p < hi; ++p) {
        mz_out *o = &q->outs[p];
        const mz_res_rec *r = &q->rec[p];
        const mz_job *j = &q->jobs[p];
        /* (every field gets its final value in ONE store, and o->block is not this loop's: chunk_finish() hangs the chunk's block on its
         *  first pair when the loop is complete, and a piece that is run a second time may still be at work then -- mz_pool.c) */
        o->status = r->status; o->badrow = r->badrow;
        if (r->status != MZ_OK) {
            const int emit = r->status == MZ_E_EMIT;        /* i, j of the reference's message */
            o->OM = emit ? r->om : 0; o->cols = NULL;
            o->score[0] = emit ? r->f[0] : 0; o->score[1] = emit ? r->f[1] : 0; o->score[2] = 0;
            continue;
        }
        o->OM = r->om;
        o->score[0] = r->f[0]; o->score[1] = r->f[1]; o->score[2] = r->f[2];
        o->cols = q->block + q->where[p];
        if (p + 1 < hi) {                                    /* (as in pack_range: the next pair's sources) */
            int k;
            for (k = 0; k < 256; k += 64) { _mm_prefetch((const char *)q->jobs[p + 1].A + k, _MM_HINT_T0); _mm_prefetch((const char *)q->jobs[p + 1].B + k, _MM_HINT_T0); }
        }
        mz_assemble_cols(j->K, j->L, j->M, j->N, j->A, j->B, q->packed + r->off, r->om, o->cols);
    }
    _mm_sfence();
}
static void results_count(int n, asm_ctx *q)               /* pairs without a result, band cells */
{
    int p;
    q->failed = 0; q->cells = 0;
    for (p = 0; p < n; ++p) { q->failed += q->rec[p].status != MZ_OK; q->cells += q->rec[p].cells; }
}

/* Does the 2-bit script of `om` columns take exactly M columns of A and N of B?  (C = 0 takes both, I = 1 one of B, D = 2 one of A;
 * 3 is no operation of the reference, mz_yama.c:24-26.)  The device checks its own scripts (MZ_E_EMIT); an image that arrives from
 * somewhere else -- another rank's, over the wire -- is only as good as this. */
static int script_fits(const uint8_t *s, int om, int M, int N)
{
    long a = 0, b = 0;
    int m;
    for (m = 0; m < om; ++m) {
        const unsigned op = (s[m >> 2] >> (2 * (m & 3))) & 3u;
        if (op == 3u) return 0;
        a += op != 1u; b += op != 2u;
    }
    return a == M && b == N;
}

/* outs[] of n jobs from their result image in host memory (64-byte header, a record per pair, the packed scripts): ONE block for the
 * merged columns, assembled on the pool threads from the caller's own A and B.  `limit`: bytes of the image when it came from
 * somewhere else (`foreign`: a link image -- every record AND every script is checked against the jobs; an image of no bytes at all
 * is too short, not "trusted"); the pipeline's own images are not `foreign` and `limit` is not looked at.
 * results_prepare(): the block and where every pair's columns go (*ac for assemble_range over [0, n)); results_close(): what hangs
 * on outs[0] afterwards; returns the failed pairs. */
static int results_prepare(int n, const mz_job *jobs, mz_out *outs, const char *r, int foreign, size_t limit, asm_ctx *ac, size_t **where_out)
{
    const mz_res_rec *rec = (const mz_res_rec *)(r + 64);
    const size_t scripts_at = 64 + mzi_al256(sizeof(mz_res_rec) * (size_t)n);
    const uint8_t *packed = (const uint8_t *)r + scripts_at;
    size_t *where, total = 0;
    uint8_t *block = NULL;
    int p;

    *where_out = NULL;
    if (foreign) {
        if (limit < scripts_at) return mzi_set_err("result image of %zu bytes is too short for %d pairs", limit, n);
        for (p = 0; p < n; ++p) {
            const mz_res_rec *q = &rec[p];
            if (q->status != MZ_OK) continue;
            if (!job_ok(&jobs[p]) || q->om < 0 || (int64_t)q->om > (int64_t)jobs[p].M + jobs[p].N || q->off < 0 ||
                scripts_at + (size_t)q->off + ((size_t)q->om + 3) / 4 > limit)
                return mzi_set_err("result image does not belong to these jobs (pair %d: %d columns at %lld)", p, q->om, (long long)q->off);
            if (!script_fits(packed + q->off, q->om, jobs[p].M, jobs[p].N))
                return mzi_set_err("result image does not belong to these jobs (pair %d: its script does not take %d columns of A and %d of B)", p, jobs[p].M, jobs[p].N);
        }
    }
    where = (size_t *)malloc(((size_t)n + 1) * sizeof *where);
    if (!where) return mzi_set_err("out of memory");
    /* ONE allocation for the chunk's merged columns; outs[first].block owns it (mz_free_outs).  Every pair's columns
     * start on a 64-byte line of their own (no line shared between two host threads; whole-line streaming stores) --
     * except the first pair's, which start the block: for a call of one pair cols IS the block, as yama() hands it on. */
    for (p = 0; p < n; ++p)
        if (rec[p].status == MZ_OK) total += (((size_t)rec[p].om * (size_t)(jobs[p].K + jobs[p].L)) + 63) & ~(size_t)63;
    if (total) {
        size_t at = 0, mis;
        block = (uint8_t *)mzi_block_get(total + 128);
        if (!block) { free(where); return mzi_set_err("out of memory for the output columns (%zu bytes)", total); }
        mis = (size_t)((uintptr_t)block & 63);
        for (p = 0; p < n; ++p) {
            where[p] = at;
            if (rec[p].status != MZ_OK) continue;
            at += (size_t)rec[p].om * (size_t)(jobs[p].K + jobs[p].L);
            at = ((at + mis + 63) & ~(size_t)63) - mis;      /* the next pair's first byte at a multiple of 64 */
        }
    }
    ac->jobs = jobs; ac->outs = outs; ac->rec = rec; ac->packed = packed; ac->where = where; ac->block = block; ac->failed = 0; ac->cells = 0;
    *where_out = where;
    return 0;
}

static int results_assemble(int n, const mz_job *jobs, mz_out *outs, const char *r, int foreign, size_t limit, int64_t *cells)
{
    asm_ctx ac;
    size_t *where;
    if (results_prepare(n, jobs, outs, r, foreign, limit, &ac, &where) < 0) return -1;
    mzi_parallel_for(n, pack_grain(n), assemble_range, &ac);
    results_count(n, &ac);
    outs[0].block = ac.block;
    *cells = ac.cells;
    free(where);
    return ac.failed;
}

/* stage 3: wait for the chunk's last kernel (the results are in host memory then), lay the merged columns' block out; the
 * assembling itself is the loop *post */
static int chunk_collect(chunk *c, mz_ajob *post)
{
    mz_ctx *X = c->X;
    const int set = c->set;

    c->t_col0 = mzi_now_s();
    HIPCK(hipEventSynchronize(X->bdone[set]));
    c->t_col1 = mzi_now_s();
    if (!c->ac && !(c->ac = (asm_ctx *)malloc(sizeof *c->ac))) return mzi_set_err("out of memory");
    free(c->where); c->where = NULL;                     /* (of the chunk that had this set before: nobody reads it any more, mz_flow.c) */
    if (results_prepare(c->n, c->jobs, c->outs, (const char *)X->h_res[set].p, 0, 0, c->ac, &c->where) < 0) return -1;
    post->fn = assemble_range; post->ctx = c->ac; post->n = c->n; post->grain = pack_grain(c->n); post->hedge = 1;
    return 0;
}

static int chunk_finish(chunk *c)
{
    results_count(c->n, c->ac);
    c->outs[0].block = c->ac->block;
    c->cells = c->ac->cells;
    c->t_col2 = mzi_now_s();
    return c->ac->failed;
}

/* ------------------------------------------------------------------------------------------------ the pipeline (mz_flow.c) */

typedef struct batch_stats { int64_t cells, bytes_up, bytes_down; } batch_stats;
static int g_last_hedged;                  /* pieces of the last call's packing / assembling loops that were run twice (mz_pool.c) */

typedef struct ypipe {
    mz_ctx *X;
    int n, up, max_pairs, threaded, slots;   /* slots: DP streams of this call (2; 4 for a call of few long pairs) */
    size_t max_bytes;
    const mz_job *jobs;
    mz_out *outs;
    chunk ck[MZ_SETS];
    batch_stats st;
    pthread_mutex_t mu;                    /* st, the report lines */
    double t0;
    hipEvent_t ev0;                        /* MZ_TIMING=2: recorded on the context's stream when the call starts */
} ypipe;

static void chunk_report(const ypipe *P, const chunk *c)
{
    float g[5] = { 0, 0, 0, 0, 0 }, since0 = 0;
    int k;
    if (g_timing < 2) return;
    if (c->X->btime_ready) {
        /* GPU time stamps of this chunk's stream: staging block -> device + expansion, plan, (wait for the host) prep + DP, walk + script
         * packing, results -> host; `gpu_start_ms`: the chunk's first stamp against the call's */
        hipSetDevice(c->X->device);
        for (k = 0; k < 5; ++k) hipEventElapsedTime(&g[k], c->X->btime[c->set][k], c->X->btime[c->set][k + 1]);
        hipEventElapsedTime(&since0, P->ev0, c->X->btime[c->set][0]);
    }
    fprintf(stderr, "{\"mz_yama_batch_chunk\": %d, \"pairs\": %d, \"stream\": %d, \"cells\": %lld, \"bytes_up\": %lld, \"bytes_down\": %lld, "
                    "\"cut_ms\": [%.3f, %.3f], \"packed_ms\": %.3f, \"sent_ms\": %.3f, \"plan_wait_ms\": [%.3f, %.3f], \"launched_ms\": %.3f, \"result_wait_ms\": [%.3f, %.3f], \"assembled_ms\": %.3f, "
                    "\"gpu_start_ms\": %.3f, \"gpu_ms\": {\"h2d_expand\": %.3f, \"plan\": %.3f, \"host_gap_dp\": %.3f, \"walk_pack\": %.3f, \"d2h\": %.3f}}\n",
            c->index, c->n, c->lane, (long long)c->cells, (long long)(c->in_bytes + c->exc_bytes), (long long)c->res_bytes,
            1e3 * (c->t_cut0 - P->t0), 1e3 * (c->t_cut1 - P->t0), 1e3 * (c->t_packed - P->t0), 1e3 * (c->t_send1 - P->t0), 1e3 * (c->t_launch0 - P->t0), 1e3 * (c->t_launch1 - P->t0),
            1e3 * (c->t_launch2 - P->t0), 1e3 * (c->t_col0 - P->t0), 1e3 * (c->t_col1 - P->t0), 1e3 * (c->t_col2 - P->t0),
            since0, g[0], g[1], g[2], g[3], g[4]);
}

/* what a pair weighs in the cutting of chunks: its input bytes as the caller holds them */
static size_t job_bytes(const mz_job *j)
{
    return job_ok(j) ? (size_t)j->K * j->M + (size_t)j->L * j->N + 8 * ((size_t)j->M + 1) : 0;
}

/* the next chunk: at most `limit` pairs and `max_bytes` of input -- but at least MIN_CHUNK_PAIRS pairs (while the hard
 * 1 GB bound allows).  A chunk's DP launch is a wave per pair and, below a wave per SIMD, takes as long as its longest
 * pair whatever their number; the chunks of a call of few long pairs only stagger those launches (and the GPU deals
 * the waves of a launch out from its first CUs on: several small launches in flight pile onto the same CUs).  C5,
 * 1 000 pairs of 100 000 rows: 12 chunks of 84 pairs 125 ms, 4 of 256 pairs 83 ms, one chunk 60 ms. */
#define MIN_CHUNK_PAIRS 1024
static int next_chunk(const mz_job *jobs, int n, int first, int limit, size_t max_bytes, int min_pairs)
{
    size_t bytes = 0;
    int m = 0;
    while (first + m < n && m < limit && (bytes < max_bytes || (m < min_pairs && bytes < ((size_t)1 << 30)))) {
        bytes += job_bytes(&jobs[first + m]);
        ++m;
    }
    return m;
}

static int y_cut(void *self, int k, int set, mz_ajob *pack)
{
    ypipe *P = (ypipe *)self;
    /* the first chunks are smaller: the GPU starts that much earlier (MZ_RAMP: their sizes in eighths of a chunk, e.g. "2,4") */
    static int ramp_n = -1, ramp8[16];
    int limit = P->max_pairs, m, shift8 = 8;
    if (ramp_n < 0) {
        const char *e = getenv("MZ_RAMP");
        ramp_n = 0;
        for (e = e ? e : "2,4"; *e && ramp_n < 16; ) { ramp8[ramp_n++] = atoi(e); while (*e && *e != ',') ++e; if (*e) ++e; }
    }
    /* (not for a call on four DP slots: its chunks' DPs take one long pair's time each whatever their size, and all of them run side by side) */
    if (P->threaded && P->slots <= 2 && k < ramp_n && (P->max_pairs >= 2048 || P->max_bytes >= ((size_t)32 << 20)) && ramp8[k] >= 1 && ramp8[k] <= 32) shift8 = ramp8[k];
    limit = (int)((long long)P->max_pairs * shift8 / 8);
    if (limit < MIN_CHUNK_PAIRS / 2) limit = MIN_CHUNK_PAIRS / 2;
    if (P->up >= P->n) return 0;
    /* (four DP slots: the call in four chunks of as many pairs each, however few -- all of them run side by side, each as long as one pair takes) */
    m = P->slots > 2 ? next_chunk(P->jobs, P->n, P->up, (P->n + P->slots - 1) / P->slots, (size_t)1 << 30, 0)
                     : next_chunk(P->jobs, P->n, P->up, limit, P->max_bytes / 8 * (size_t)shift8, MIN_CHUNK_PAIRS);
    if (chunk_cut(P->X, &P->ck[set], k, set, P->threaded ? (k % P->slots) | (P->slots > 2 ? 0x100 | (P->n <= 1024 ? 0x200 : 0) : 0) : -1, m, P->jobs + P->up, P->outs + P->up, pack) < 0) return -1;
    P->up += m;
    return 1;
}
static int y_send(void *self, int k, int set, mz_ajob *post) { (void)k; (void)post; return chunk_send(&((ypipe *)self)->ck[set]); }
static int y_launch(void *self, int k, int set, mz_ajob *post) { (void)k; (void)post; return chunk_launch(&((ypipe *)self)->ck[set]); }
static int y_collect(void *self, int k, int set, mz_ajob *post) { (void)k; return chunk_collect(&((ypipe *)self)->ck[set], post); }
static int y_finish(void *self, int k, int set)
{
    ypipe *P = (ypipe *)self;
    chunk *c = &P->ck[set];
    const int failed = chunk_finish(c);
    (void)k;
    pthread_mutex_lock(&P->mu);
    chunk_report(P, c);
    P->st.cells += c->cells; P->st.bytes_up += c->in_bytes + c->exc_bytes; P->st.bytes_down += c->res_bytes;
    pthread_mutex_unlock(&P->mu);
    return failed;
}

/* the stage threads of a context are persistent (mz_pool.c: mzi_workers_start / mzi_worker_give) */
void mzi_workers_stop(mz_ctx *X)
{
    mzi_workers_end(X->fworker, MZ_FLOW_STAGES);
}

/* A batch of any size on ONE context.  A guide-tree level with a million merges (BASELINE config 4) needs a bounded
 * amount of pinned host memory and HBM: MZ_SETS chunks at most are in flight.  On a device error everything in flight
 * is drained and every pair not yet collected is left marked MZ_E_DEVICE with cols == NULL (mz_yama_batch() pre-marks
 * all of them), so a caller may clean up outs. */
static int batch_on_ctx(mz_ctx *X, int n, const mz_job *jobs, mz_out *outs, int max_pairs, batch_stats *stats)
{
    ypipe *P;
    mz_flow *F;
    size_t max_bytes = 0, total_bytes = 0;
    int rc, s;

    if (hipSetDevice(X->device) != hipSuccess) return mzi_set_err("hipSetDevice(%d) failed", X->device);
    P = (ypipe *)calloc(1, sizeof *P);
    F = (mz_flow *)calloc(1, sizeof *F);
    if (!P || !F) { free(P); free(F); return mzi_set_err("out of memory"); }
    P->X = X; P->n = n; P->jobs = jobs; P->outs = outs; P->max_pairs = max_pairs; P->t0 = mzi_now_s();
    pthread_mutex_init(&P->mu, NULL);

    if (g_timing >= 2) {
        if (!X->btime_ready) {
            int a, e, ok = 1;
            for (a = 0; a < MZ_SETS && ok; ++a) for (e = 0; e < 6 && ok; ++e) ok = hipEventCreate(&X->btime[a][e]) == hipSuccess;
            X->btime_ready = ok;
        }
        if (X->btime_ready && hipEventCreate(&P->ev0) == hipSuccess) hipEventRecord(P->ev0, X->stream);
        else X->btime_ready = 0;
    }
    /* chunks by pairs AND by bytes: a call of few long pairs (BASELINE config 5: 1 000 pairs, 1.2 GB) is cut into as many
     * pieces as one of many short ones, at least 8 MB each and at most 1 GB */
    for (s = 0; s < n; ++s) max_bytes += job_bytes(&jobs[s]);
    total_bytes = max_bytes;
    max_bytes = max_bytes / (size_t)chunk_parts(n) + 1;
    if (max_bytes < ((size_t)8 << 20)) max_bytes = (size_t)8 << 20;
    if (max_bytes > ((size_t)1 << 30)) max_bytes = (size_t)1 << 30;
    P->max_bytes = max_bytes;
    /* one chunk: the steps inline (no thread is woken for a single yama() call); also when no thread can be had */
    P->threaded = next_chunk(jobs, n, 0, max_pairs, max_bytes, MIN_CHUNK_PAIRS) < n;
    F->X = X; F->self = P; F->nstage = 3; F->threaded = P->threaded;
    F->cut = y_cut; F->stage[0] = y_send; F->stage[1] = y_launch; F->stage[2] = y_collect; F->finish = y_finish;
    if (P->threaded && mzi_flow_streams(X)) { P->threaded = F->threaded = 0; }
    P->slots = 2;
    /* few long pairs (at most 4 096 pairs, cut by bytes into three chunks or more): every chunk's DP takes as long as its longest pair --
     * milliseconds -- on a fraction of the GPU's SIMDs: four of them side by side (mz_flow.c: mzi_flow_wide; MZ_WIDE=0: two, as before) */
    {
        static int wide_on = -1;
        if (wide_on < 0) { const char *e = getenv("MZ_WIDE"); wide_on = !(e && e[0] == '0'); }
        if (P->threaded && wide_on && n <= 4096 && total_bytes / (size_t)n >= ((size_t)256 << 10) && mzi_flow_wide(X) == 0) P->slots = MZ_QS;      /* (long: a quarter of a megabyte of input a pair and more) */
    }
    rc = mzi_flow_run(F);
    if (rc < 0) mzi_flow_sync(X);
    g_last_hedged = F->hedged;
    for (s = 0; s < MZ_SETS; ++s) { free((void *)P->ck[s].pc.slotC); free(P->ck[s].ac); free(P->ck[s].where); }
    if (g_timing >= 2 && X->btime_ready) hipEventDestroy(P->ev0);
    if (stats) *stats = P->st;
    pthread_mutex_destroy(&P->mu);
    free(P); free(F);
    return rc;
}

/* ------------------------------------------------------------------------------------------------ several GPUs */

/* host thread of one further GPU (mz_yama_batch with several contexts) */
typedef struct dev_task { mz_ctx *X; int n, max_pairs, rc; const mz_job *jobs; mz_out *outs; batch_stats st; char err[600]; } dev_task;
static void *dev_worker(void *arg)
{
    dev_task *t = (dev_task *)arg;
    t->rc = batch_on_ctx(t->X, t->n, t->jobs, t->outs, t->max_pairs, &t->st);
    if (t->rc < 0) snprintf(t->err, sizeof t->err, "GPU %d: %s", t->X->device, mz_last_error());
    return NULL;
}

/* what a pair costs the GPU, roughly: band rows x the band's width in the middle (no pass over the bounds) */
static double job_weight(const mz_job *j)
{
    if (j->K < 1 || j->L < 1 || j->M < 1 || j->N < 1 || !j->LB || !j->RB) return 1.0;
    return ((double)j->M + 1.0) * (double)(j->RB[j->M / 2] - j->LB[j->M / 2] + 1) + 64.0 * (j->K + j->L);
}

/* bytes the last mz_yama_batch() call moved over the link, each way */
static int64_t g_last_up, g_last_down;

/* Work dealt over `use` GPUs by cost: the items are classed by weight in half octaves (a counting sort: no comparison sort of a
 * million-item list) and dealt out from the heaviest class down in a snake (0..G-1, G-1..0, ...), as multiz_amd/shard.py deals a list
 * over ranks: every GPU gets the same MIX -- a list that arrives long-pairs-first no longer gives the first GPU the few long pairs
 * and the last all the short ones (contiguous ranges balance the total weight only).  owner[i]: the item's GPU; where[i]: its place in
 * the GPU-major order (GPU d's items are [start[d], start[d] + cnt[d]), in their original order). */
int mzi_deal_snake(int n, const double *weight, int use, int *owner, int *where, int *cnt, int *start)
{
    unsigned char *cls = (unsigned char *)malloc((size_t)(n ? n : 1));
    int ccount[64], cstart[64], pos, c, d, p;
    if (!cls) return mzi_set_err("out of memory");
    if (use < 1) { free(cls); return mzi_set_err("mzi_deal_snake: nobody to deal to"); }
    memset(ccount, 0, sizeof ccount);
    for (p = 0; p < n; ++p) {                                /* half-octave class of the weight: 2 * log2 */
        int e = 0;
        const double m = frexp(weight[p] > 1.0 ? weight[p] : 1.0, &e);       /* w = m * 2^e, m in [0.5, 1) */
        c = 2 * e + (m >= 0.70710678 ? 1 : 0);
        cls[p] = (unsigned char)(c < 0 ? 0 : c > 63 ? 63 : c);
        ccount[cls[p]]++;
    }
    for (c = 63, pos = 0; c >= 0; --c) { cstart[c] = pos; pos += ccount[c]; }     /* heaviest class first */
    for (d = 0; d < use; ++d) cnt[d] = 0;
    for (p = 0; p < n; ++p) {                                /* rank of the item in the sorted order -> its GPU, in a snake */
        const int rank = cstart[cls[p]]++, rnd = rank / use, k = rank % use;
        owner[p] = (rnd & 1) ? use - 1 - k : k;
        cnt[owner[p]]++;
    }
    /* (`use` is a rank count when mz_shard_scatter() deals -- any number, not at most MZ_MAX_DEV GPUs of one process: start[] itself is
     *  the running fill pointer and is put back afterwards; a fixed array of MZ_MAX_DEV on the stack was overrun by a world of 17) */
    for (d = 0, pos = 0; d < use; ++d) { start[d] = pos; pos += cnt[d]; }
    for (p = 0; p < n; ++p) where[p] = start[owner[p]]++;
    for (d = 0; d < use; ++d) start[d] -= cnt[d];
    free(cls);
    return 0;
}

/* A planned link image lives in buffer set 0 of the primary context (mz_link_plan): whatever else runs there takes it away, and
 * mz_link_finish() then refuses instead of aligning somebody else's pools.  Called with g_big held. */
static int g_link_planned;
void mzi_link_forget(void) { g_link_planned = 0; }

int mz_yama_batch(int n, const mz_job *jobs, mz_out *outs)
{
    static int env_pairs = -1;
    int failed = 0, max_pairs, use, p, rc;
    batch_stats st = { 0, 0, 0 };
    double t_call = mzi_now_s();
    if (env_pairs < 0) { const char *e = getenv("MZ_CHUNK_PAIRS"); env_pairs = e && atoi(e) > 0 ? atoi(e) : 0; }
    if (g_timing < 0) g_timing = mzi_timing();
    if (n <= 0) return 0;
    for (p = 0; outs && p < n; ++p) {                    /* "not computed" until a chunk says otherwise */
        outs[p].status = MZ_E_DEVICE; outs[p].badrow = -1; outs[p].OM = 0; outs[p].cols = NULL; outs[p].block = NULL;
        outs[p].score[0] = outs[p].score[1] = outs[p].score[2] = 0;
    }
    pthread_mutex_lock(&g_big);
    mzi_link_forget();
    {
        const int first = !g_ndev && g_timing;
        double t0 = mzi_now_s(), t1;
        if (mzi_ensure_init()) { pthread_mutex_unlock(&g_big); return -1; }
        t1 = mzi_now_s();
        if (mzi_sync_scores()) { pthread_mutex_unlock(&g_big); return -1; }
        if (!jobs || !outs) { pthread_mutex_unlock(&g_big); return mzi_set_err("mz_yama_batch: NULL jobs or outs"); }
        if (first) fprintf(stderr, "{\"mz_start_up\": {\"hip_runtime_and_streams_ms\": %.1f, \"gpus\": %d, \"score_upload_and_code_object_ms\": %.1f}}\n",
                           1e3 * (t1 - t0), g_ndev, 1e3 * (mzi_now_s() - t1));
    }
    /* GPUs to use: all of them once every one gets a worthwhile share */
    use = g_ndev;
    while (use > 1 && n / use < MZ_MULTI_MIN) --use;
    {
        /* chunk size: a GPU's share in chunk_parts() pieces, so that the copies, the kernels and the host's packing and assembling of
         * different chunks overlap and the first kernels start early -- but at least 1 Ki pairs (a wave per SIMD; chunks in flight
         * share the GPU) and at most 16 Ki */
        const int share = (n + use - 1) / use, parts = chunk_parts(share), per = (share + parts - 1) / parts;
        max_pairs = env_pairs ? env_pairs : per < 1024 ? 1024 : per > 16384 ? 16384 : per;
    }
    if (use == 1) {
        G.copy_threads = MZ_COPY_THREADS;
        rc = batch_on_ctx(&G, n, jobs, outs, max_pairs, &st);
    } else {
        /* Dealt by cost, one share per GPU, each driven by its own host thread through its own context (streams, staging
         * buffers, helper threads).  The pairs are classed by weight in half octaves (a counting sort: no comparison sort of a
         * million-pair list) and dealt out from the heaviest class down in a snake over the GPUs (0..G-1, G-1..0, ...), as
         * multiz_amd/shard.py deals a list over ranks: every GPU gets the same MIX -- a list that arrives long-pairs-first no
         * longer gives the first GPU the few long pairs and the last all the short ones (contiguous ranges balanced the total
         * weight only).  Each GPU's jobs are copied into a list of its own (48 bytes a job), its results copied back to the
         * jobs' own positions.  No data-path collective: the work list lives in host memory and every GPU pulls its share
         * over its own PCIe link. */
        dev_task task[MZ_MAX_DEV];
        pthread_t th[MZ_MAX_DEV];
        int d, started[MZ_MAX_DEV], cnt[MZ_MAX_DEV], first[MZ_MAX_DEV];
        int *owner = (int *)malloc((size_t)n * sizeof *owner), *where = (int *)malloc((size_t)n * sizeof *where);
        double *wt = (double *)malloc((size_t)n * sizeof *wt);
        mz_job *jbuf = (mz_job *)malloc((size_t)n * sizeof *jbuf);
        mz_out *obuf = (mz_out *)malloc((size_t)n * sizeof *obuf);
        if (!owner || !where || !wt || !jbuf || !obuf) {
            free(owner); free(where); free(wt); free(jbuf); free(obuf);
            pthread_mutex_unlock(&g_big);
            return mzi_set_err("out of memory");
        }
        for (p = 0; p < n; ++p) wt[p] = job_weight(&jobs[p]);
        if (mzi_deal_snake(n, wt, use, owner, where, cnt, first)) { free(owner); free(where); free(wt); free(jbuf); free(obuf); pthread_mutex_unlock(&g_big); return -1; }
        free(wt);
        for (d = 0; d < use; ++d) {
            task[d].X = &g_dev[d]; task[d].jobs = jbuf + first[d]; task[d].outs = obuf + first[d];
            task[d].n = cnt[d]; task[d].max_pairs = max_pairs; task[d].rc = 0; task[d].err[0] = 0; memset(&task[d].st, 0, sizeof task[d].st);
            g_dev[d].copy_threads = MZ_COPY_THREADS * 2 / use < 4 ? 4 : MZ_COPY_THREADS * 2 / use > MZ_COPY_THREADS ? MZ_COPY_THREADS : MZ_COPY_THREADS * 2 / use;
        }
        for (p = 0; p < n; ++p) { jbuf[where[p]] = jobs[p]; obuf[where[p]] = outs[p]; }
        for (d = 1; d < use; ++d) {
            started[d] = task[d].n > 0 && pthread_create(&th[d], NULL, dev_worker, &task[d]) == 0;
            if (!started[d] && task[d].n > 0) dev_worker(&task[d]);         /* no thread: do it here, after the others started */
        }
        if (task[0].n > 0) dev_worker(&task[0]);
        rc = 0;
        for (d = 0; d < use; ++d) {
            if (d >= 1 && started[d]) pthread_join(th[d], NULL);
            if (task[d].n <= 0) continue;
            if (task[d].rc < 0) { rc = -1; mzi_set_err("%s", task[d].err); }
            else failed += task[d].rc;
            st.cells += task[d].st.cells; st.bytes_up += task[d].st.bytes_up; st.bytes_down += task[d].st.bytes_down;
            g_dev[d].copy_threads = MZ_COPY_THREADS;
        }
        for (p = 0; p < n; ++p) outs[p] = obuf[where[p]];    /* (a chunk's result block hangs on its first pair: mz_free_outs() walks all n) */
        free(owner); free(where); free(jbuf); free(obuf);
        hipSetDevice(G.device);
        if (rc >= 0) rc = failed;
    }
    g_last_up = st.bytes_up; g_last_down = st.bytes_down;
    pthread_mutex_unlock(&g_big);
    if (g_timing && rc >= 0 && !mzi_warm_thread) {
        /* one JSON line per call: pairs, band cells, seconds, GCUPS, bytes over the link each way */
        const double dt = mzi_now_s() - t_call;
        fprintf(stderr, "{\"mz_yama_batch\": {\"pairs\": %d, \"failed\": %d, \"cells\": %lld, \"seconds\": %.6f, \"gcups\": %.2f, \"bytes_up\": %lld, \"bytes_down\": %lld, "
                        "\"gpus\": %d, \"chunk_pairs\": %d, \"pieces_run_twice\": %d}}\n", n, rc, (long long)st.cells, dt, (double)st.cells / dt / 1e9, (long long)st.bytes_up, (long long)st.bytes_down, use, max_pairs, g_last_hedged);
    }
    return rc;
}

/* ------------------------------------------------------------------------------------------------ link images
 * (include/mz_amd.h: the traffic of a chunk as something the caller moves -- multiz_amd/shard.py sends it over RCCL).  The image is
 * the staging block of chunk_upload(): [K L M N](int32 x n) [offA offB offBand](int64 x n) [bandLen LB0 RB0](int32 x n) offC(int64 x n)
 * fmt(n) band steps, class nibbles of A, class nibbles of B -- every part at a multiple of 256 bytes. */
typedef struct link_parts { size_t K, L, M, N, offA, offB, offBand, len, lb0, rb0, offC, fmt, steps, nibA, nibB, bytes; } link_parts;

static int link_lay(const mz_link_desc *d, link_parts *y)
{
    size_t at = 0;
    const size_t n = (size_t)d->n;
    if (d->n < 0 || d->n > INT32_MAX || d->colsA < 0 || d->colsB < 0 || d->band < 0 || d->steps < 0 || d->exc_bytes < 0 || (d->colsA & 63) || (d->colsB & 63))
        return mzi_set_err("not a link descriptor");
#define PART(f, bytes) do { y->f = at; at += mzi_al256(bytes); } while (0)
    PART(K, 4 * n); PART(L, 4 * n); PART(M, 4 * n); PART(N, 4 * n);
    PART(offA, 8 * n); PART(offB, 8 * n); PART(offBand, 8 * n);
    PART(len, 4 * n); PART(lb0, 4 * n); PART(rb0, 4 * n); PART(offC, 8 * n); PART(fmt, n);
    PART(steps, (size_t)d->steps); PART(nibA, (size_t)d->colsA / 2); PART(nibB, (size_t)d->colsB / 2);
#undef PART
    y->bytes = at;
    return 0;
}

static void *link_alloc(size_t bytes)
{
    void *p = NULL;
    return posix_memalign(&p, 256, bytes ? mzi_al256(bytes) : 256) ? NULL : p;
}
void mz_link_free(void *p) { free(p); }

int mz_link_pack(int n, const mz_job *jobs, mz_link_desc *d, void **image, void **exc)
{
    link_parts y;
    size_t eA = 0, eB = 0, nband = 0, bytesC = 0, oa = 0, ob = 0, oband = 0, oc = 0, bytesE = 0;
    int32_t *hK, *hL, *hM, *hN, *hLen, *hLB0, *hRB0;
    int64_t *hoA, *hoB, *hoBand, *hoC;
    uint32_t *esz;
    char *h;
    pack_ctx pc;
    int p;

    if (image) *image = NULL;
    if (exc) *exc = NULL;
    if (n < 0 || (n && !jobs) || !d || !image || !exc) return mzi_set_err("mz_link_pack: bad arguments");
    for (p = 0; p < n; ++p) {
        const mz_job *j = &jobs[p];
        if (job_ok(j)) { eA += cols_padded(j->K, j->M); eB += cols_padded(j->L, j->N); nband += (size_t)j->M + 1; bytesC += band_slot(j->M); }
        else nband += 1;
    }
    memset(d, 0, sizeof *d);
    d->n = n; d->colsA = (int64_t)eA; d->colsB = (int64_t)eB; d->band = (int64_t)nband; d->steps = (int64_t)bytesC;
    if (link_lay(d, &y)) return -1;
    h = (char *)link_alloc(y.bytes);
    esz = (uint32_t *)malloc(((size_t)n + 1) * sizeof *esz);
    if (!h || !esz) { free(h); free(esz); return mzi_set_err("out of memory"); }
    memset(h, 0, y.steps);                                   /* (the gaps between the header's parts travel too ... */
    memset(h + y.steps + bytesC, 0, y.nibA - (y.steps + bytesC));                      /* ... and those behind the steps and the nibbles) */
    memset(h + y.nibA + eA / 2, 0, y.nibB - (y.nibA + eA / 2));
    memset(h + y.nibB + eB / 2, 0, y.bytes - (y.nibB + eB / 2));
    hK = (int32_t *)(h + y.K); hL = (int32_t *)(h + y.L); hM = (int32_t *)(h + y.M); hN = (int32_t *)(h + y.N);
    hoA = (int64_t *)(h + y.offA); hoB = (int64_t *)(h + y.offB); hoBand = (int64_t *)(h + y.offBand);
    hLen = (int32_t *)(h + y.len); hLB0 = (int32_t *)(h + y.lb0); hRB0 = (int32_t *)(h + y.rb0); hoC = (int64_t *)(h + y.offC);
    for (p = 0; p < n; ++p) {
        const mz_job *j = &jobs[p];
        const int ok = job_ok(j);
        const int nul = !j->A || !j->B || !j->LB || !j->RB;
        hK[p] = j->K; hL[p] = j->L; hM[p] = nul ? 0 : j->M; hN[p] = nul ? 0 : j->N;
        hoA[p] = (int64_t)oa; hoB[p] = (int64_t)ob; hoBand[p] = (int64_t)oband; hoC[p] = (int64_t)oc;
        hLen[p] = ok ? j->M + 1 : 1;
        hLB0[p] = ok ? j->LB[0] : 0; hRB0[p] = ok ? j->RB[0] : 0;
        if (ok) { oa += cols_padded(j->K, j->M); ob += cols_padded(j->L, j->N); oband += (size_t)j->M + 1; oc += band_slot(j->M); }
        else oband += 1;
    }
    pc.jobs = jobs; pc.hoA = hoA; pc.hoB = hoB; pc.hoC = hoC; pc.hA = (uint8_t *)h + y.nibA; pc.hB = (uint8_t *)h + y.nibB;
    pc.hC = (uint8_t *)h + y.steps; pc.hFmt = (uint8_t *)h + y.fmt; pc.esz = esz; pc.hE = NULL; pc.hLB0 = pc.hRB0 = NULL; pc.slotC = NULL;
    mzi_parallel_for(n, pack_grain(n), pack_range, &pc);
    for (p = 0; p < n; ++p) if (esz[p]) { hoC[p] = (int64_t)bytesE; bytesE += esz[p]; }
    if (bytesE) {
        uint8_t *hE = (uint8_t *)link_alloc(bytesE);
        if (!hE) { free(h); free(esz); return mzi_set_err("out of memory"); }
        pc.hE = hE;
        mzi_parallel_for(n, pack_grain(n), pack_exceptions, &pc);
        *exc = hE;
    }
    free(esz);
    d->image_bytes = (int64_t)y.bytes; d->exc_bytes = (int64_t)bytesE;
    *image = h;
    return 0;
}

int mz_link_parts(const mz_link_desc *d, int64_t at[16])
{
    link_parts y;
    if (!d || !at || link_lay(d, &y)) return -1;
    at[0] = (int64_t)y.K; at[1] = (int64_t)y.L; at[2] = (int64_t)y.M; at[3] = (int64_t)y.N; at[4] = (int64_t)y.offA; at[5] = (int64_t)y.offB;
    at[6] = (int64_t)y.offBand; at[7] = (int64_t)y.len; at[8] = (int64_t)y.lb0; at[9] = (int64_t)y.rb0; at[10] = (int64_t)y.offC; at[11] = (int64_t)y.fmt;
    at[12] = (int64_t)y.steps; at[13] = (int64_t)y.nibA; at[14] = (int64_t)y.nibB; at[15] = (int64_t)y.bytes;
    return 0;
}

int mz_link_expand(const mz_link_desc *d, const void *dev_image, const void *dev_exc, void *dev_cols, void *dev_LB, void *dev_RB, void *stream)
{
    link_parts y;
    const char *di = (const char *)dev_image;
    hipStream_t st;
    if (!d || !dev_image || !dev_cols || !dev_LB || !dev_RB || (d->exc_bytes && !dev_exc)) return mzi_set_err("mz_link_expand: bad arguments");
    if (link_lay(d, &y)) return -1;
    if ((int64_t)y.bytes != d->image_bytes) return mzi_set_err("mz_link_expand: not this image's descriptor");
    pthread_mutex_lock(&g_big);
    if (mzi_ensure_init()) { pthread_mutex_unlock(&g_big); return -1; }
    st = stream ? (hipStream_t)stream : G.stream;
    if (mzk_unband((int)d->n, (const int32_t *)(di + y.len), (const int32_t *)(di + y.lb0), (const int32_t *)(di + y.rb0), (const int64_t *)(di + y.offBand),
                   (const int64_t *)(di + y.offC), (const uint8_t *)(di + y.fmt), (const uint8_t *)(di + y.steps), (const uint8_t *)dev_exc,
                   (int32_t *)dev_LB, (int32_t *)dev_RB, st) ||
        mzk_unnib(di + y.nibA, dev_cols, (long long)(2 * (mzi_al256((size_t)d->colsA / 2) + mzi_al256((size_t)d->colsB / 2))), st)) {
        mzi_set_err("%s", mzk_last_error()); pthread_mutex_unlock(&g_big); return -1;
    }
    pthread_mutex_unlock(&g_big);
    return 0;
}

/* the planned image between mz_link_plan() and mz_link_finish() */
static struct { mz_dev_batch b; int64_t n, res_bytes, totals[16]; int planned; } g_link;

int mz_link_plan(mz_link_desc *d, const void *dev_image, const void *dev_exc, void *stream)
{
    mz_ctx *X = &G;
    link_parts y;
    mz_dev_batch b;
    hipStream_t st;
    const char *di = (const char *)dev_image;
    const int64_t *totals;
    int n;

    if (!d || (d->n && !dev_image) || (d->exc_bytes && !dev_exc)) return mzi_set_err("mz_link_plan: bad arguments");
    if (link_lay(d, &y)) return -1;
    if ((int64_t)y.bytes != d->image_bytes) return mzi_set_err("mz_link_plan: the descriptor says %lld bytes, its parts add up to %zu", (long long)d->image_bytes, y.bytes);
    pthread_mutex_lock(&g_big);
#define LFAIL(x) do { if (x) { pthread_mutex_unlock(&g_big); return -1; } } while (0)
    LFAIL(mzi_ensure_init() || mzi_sync_scores());
    g_link.planned = 0;
    n = (int)d->n;
    d->res_bytes = 64 + (int64_t)mzi_al256(sizeof(mz_res_rec) * (size_t)n) + 64;
    if (n == 0) { g_link.n = 0; g_link.res_bytes = d->res_bytes; g_link.planned = 1; g_link_planned = 1; pthread_mutex_unlock(&g_big); return 0; }
    st = stream ? (hipStream_t)stream : X->stream;
    LFAIL(mzi_dev_reserve(&X->d_cols[0], 2 * (mzi_al256((size_t)d->colsA / 2) + mzi_al256((size_t)d->colsB / 2)) + 256) ||
          mzi_dev_reserve(&X->d_band[0], 2 * mzi_al256(4 * (size_t)d->band)) ||
          mzi_dev_reserve(&X->d_plan[0], mz_dev_plan_bytes(n)) || mzi_host_reserve(&X->h_tot[0], 16 * sizeof(int64_t)));
    memset(&b, 0, sizeof b);
    b.n = n;
    b.K = (const int32_t *)(di + y.K); b.L = (const int32_t *)(di + y.L); b.M = (const int32_t *)(di + y.M); b.N = (const int32_t *)(di + y.N);
    b.offA = (const int64_t *)(di + y.offA); b.offB = (const int64_t *)(di + y.offB); b.offBand = (const int64_t *)(di + y.offBand);
    b.poolA = (const uint8_t *)X->d_cols[0].p;
    b.poolB = b.poolA + 2 * mzi_al256((size_t)d->colsA / 2);
    b.poolLB = (const int32_t *)X->d_band[0].p;
    b.poolRB = (const int32_t *)((char *)X->d_band[0].p + mzi_al256(4 * (size_t)d->band));
    if (mzk_unband(n, (const int32_t *)(di + y.len), (const int32_t *)(di + y.lb0), (const int32_t *)(di + y.rb0), b.offBand, (const int64_t *)(di + y.offC),
                   (const uint8_t *)(di + y.fmt), (const uint8_t *)(di + y.steps), (const uint8_t *)dev_exc, (int32_t *)b.poolLB, (int32_t *)b.poolRB, st) ||
        mzk_unnib(di + y.nibA, (void *)b.poolA, (long long)(2 * (mzi_al256((size_t)d->colsA / 2) + mzi_al256((size_t)d->colsB / 2))), st)) {
        mzi_set_err("%s", mzk_last_error()); pthread_mutex_unlock(&g_big); return -1;
    }
    mz_dev_carve(&b, X->d_plan[0].p);
    b.capTb = b.capScript = b.capOut = b.capPrep = INT64_MAX;
    if (mzk_plan(&b, st)) { mzi_set_err("%s", mzk_last_error()); pthread_mutex_unlock(&g_big); return -1; }
    if (hipMemcpyAsync(X->h_tot[0].p, b.totals, 16 * sizeof(int64_t), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
        mzi_set_err("mz_link_plan: the plan did not come back: %s", hipGetErrorString(hipGetLastError())); pthread_mutex_unlock(&g_big); return -1;
    }
    totals = (const int64_t *)X->h_tot[0].p;
    d->res_bytes = 64 + (int64_t)mzi_al256(sizeof(mz_res_rec) * (size_t)n) + totals[1] / 4 + 64;
    memcpy(g_link.totals, totals, sizeof g_link.totals);         /* (h_tot[0] is the chunk pipelines' too) */
    g_link.b = b; g_link.n = n; g_link.res_bytes = d->res_bytes; g_link.planned = 1; g_link_planned = 1;
    pthread_mutex_unlock(&g_big);
    return 0;
}

int mz_link_finish(const mz_link_desc *d, void *dev_result, void *stream)
{
    mz_ctx *X = &G;
    mz_dev_batch b;
    const int64_t *totals;
    hipStream_t st;
    char *dres = (char *)dev_result;
    int n;

    pthread_mutex_lock(&g_big);
    if (!d || !g_link.planned || !g_link_planned || d->n != g_link.n || d->res_bytes != g_link.res_bytes || !dev_result) {
        pthread_mutex_unlock(&g_big);
        return mzi_set_err("mz_link_finish: not the image mz_link_plan() planned last (or another batch call has run since)");
    }
    g_link.planned = 0;
    n = (int)d->n;
    st = stream ? (hipStream_t)stream : X->stream;
    if (n == 0) { pthread_mutex_unlock(&g_big); return 0; }
    b = g_link.b;
    totals = g_link.totals;
    LFAIL(mzi_dev_reserve(&X->d_tb[0], 4 * (size_t)totals[0] + 256) || mzi_dev_reserve(&X->d_script[0], (size_t)totals[1] + 256) ||
          mzi_dev_reserve(&X->d_prep[0], 4 * (size_t)totals[4] + 256));
    b.tbw = (uint32_t *)X->d_tb[0].p; b.script = (uint8_t *)X->d_script[0].p; b.out = NULL;
    b.prep = (uint32_t *)X->d_prep[0].p; b.capPrep = (int64_t)(X->d_prep[0].cap / 4);
    b.walk_hint = mz_walk_choice(n, totals); b.dp_hint = mz_dp_hint(n, totals);
    b.dp_grid = mz_dp_grid(n, totals); b.dp_rows = mz_dp_rows(n, totals); b.hint_gen = g_hint_gen;
    b.capTb = (int64_t)(X->d_tb[0].cap / 4); b.capScript = (int64_t)X->d_script[0].cap; b.capOut = INT64_MAX;
    if (mzk_prep(&b, st) || mzk_dp(&b, st) || mzk_walk(&b, st, 0) ||
        mzk_script_pack(&b, dres, dres + 64, dres + 64 + mzi_al256(sizeof(mz_res_rec) * (size_t)n), st)) {
        mzi_set_err("%s", mzk_last_error()); pthread_mutex_unlock(&g_big); return -1;
    }
#undef LFAIL
    pthread_mutex_unlock(&g_big);
    return 0;
}

/* the least a result image of n pairs can be: its 64-byte header and the records (the scripts follow) */
int64_t mzi_result_image_min(int n) { return n > 0 ? 64 + (int64_t)mzi_al256(sizeof(mz_res_rec) * (size_t)n) : 0; }

int mz_link_assemble(int n, const mz_job *jobs, const void *result, int64_t res_bytes, mz_out *outs)
{
    int64_t cells = 0;
    int p;
    if (n < 0 || (n && (!jobs || !outs || !result)) || res_bytes < 0) return mzi_set_err("mz_link_assemble: bad arguments");
    if (n == 0) return 0;
    for (p = 0; p < n; ++p) {
        outs[p].status = MZ_E_DEVICE; outs[p].badrow = -1; outs[p].OM = 0; outs[p].cols = NULL; outs[p].block = NULL;
        outs[p].score[0] = outs[p].score[1] = outs[p].score[2] = 0;
    }
    if ((uint64_t)res_bytes < (uint64_t)mzi_result_image_min(n)) return mzi_set_err("mz_link_assemble: a result image of %lld bytes is too short for the records of %d pairs (%lld bytes at least)", (long long)res_bytes, n, (long long)mzi_result_image_min(n));
    return results_assemble(n, jobs, outs, (const char *)result, 1, (size_t)res_bytes, &cells);
}

void mz_link_bytes(int64_t *up, int64_t *down) { if (up) *up = g_last_up; if (down) *down = g_last_down; }

/* release the merged columns of a finished mz_yama_batch() call: the blocks the chunks allocated (outs[i].block).
 * Large blocks are parked for the next calls instead of going back to the system (mz_pool.c). */
void mz_free_outs(int n, mz_out *outs)
{
    int p;
    for (p = 0; p < n; ++p) { mzi_block_put(outs[p].block); outs[p].block = NULL; outs[p].cols = NULL; }
}

/* ------------------------------------------------------------------------------------------------ host-side probe
 * tests/tools/hostprobe.py: what the packing of a chunk costs on this machine's host threads, without a GPU in the
 * loop -- the same pack_range() over the same jobs into an ordinary (not pinned) staging block, `reps` times; returns
 * seconds per repetition, or -1.  what: 1 = everything, 2 = column classes only, 3 = band steps only. */
typedef struct probe_ctx { pack_ctx pc; int what; } probe_ctx;
static void probe_range(void *ctx, int lo, int hi)
{
    const probe_ctx *P = (const probe_ctx *)ctx;
    const pack_ctx *q = &P->pc;
    int p;
    if (P->what == 1) { pack_range((void *)q, lo, hi); return; }
    for (p = lo; p < hi; ++p) {
        const mz_job *j = &q->jobs[p];
        if (!job_ok(j)) continue;
        if (P->what == 2) {
            mz_pack_classes_stream(j->A, (size_t)j->K * j->M, q->hA + q->hoA[p] / 2, cols_padded(j->K, j->M) / 2);
            mz_pack_classes_stream(j->B, (size_t)j->L * j->N, q->hB + q->hoB[p] / 2, cols_padded(j->L, j->N) / 2);
        } else {
            mz_pack_band_nib_stream(j->LB, j->RB, j->M, q->hC + q->hoC[p], band_slot(j->M));
        }
    }
    _mm_sfence();
}
double mz_host_pack_probe(int n, const mz_job *jobs, int what, int reps)
{
    size_t eA = 0, eB = 0, bytesC = 0, oa = 0, ob = 0, oc = 0;
    int64_t *hoA, *hoB, *hoC;
    uint8_t *buf, *fmt;
    uint32_t *esz;
    probe_ctx P;
    double t0;
    int p, r;
    for (p = 0; p < n; ++p) if (job_ok(&jobs[p])) { eA += cols_padded(jobs[p].K, jobs[p].M); eB += cols_padded(jobs[p].L, jobs[p].N); bytesC += band_slot(jobs[p].M); }
    hoA = (int64_t *)malloc(3 * ((size_t)n + 1) * sizeof *hoA); hoB = hoA + n + 1; hoC = hoB + n + 1;
    buf = (uint8_t *)aligned_alloc(256, mzi_al256(eA / 2) + mzi_al256(eB / 2) + mzi_al256(bytesC) + 256);
    fmt = (uint8_t *)malloc((size_t)n + 1); esz = (uint32_t *)malloc(((size_t)n + 1) * sizeof *esz);
    if (!hoA || !buf || !fmt || !esz) return -1.0;
    for (p = 0; p < n; ++p) {
        hoA[p] = (int64_t)oa; hoB[p] = (int64_t)ob; hoC[p] = (int64_t)oc;
        if (job_ok(&jobs[p])) { oa += cols_padded(jobs[p].K, jobs[p].M); ob += cols_padded(jobs[p].L, jobs[p].N); oc += band_slot(jobs[p].M); }
    }
    P.what = what;
    P.pc.jobs = jobs; P.pc.hoA = hoA; P.pc.hoB = hoB; P.pc.hoC = hoC; P.pc.hFmt = fmt; P.pc.esz = esz; P.pc.hE = NULL; P.pc.hLB0 = P.pc.hRB0 = NULL; P.pc.slotC = NULL;
    P.pc.hC = buf; P.pc.hA = buf + mzi_al256(bytesC); P.pc.hB = P.pc.hA + mzi_al256(eA / 2);
    mzi_parallel_for(n, pack_grain(n), probe_range, &P);             /* warm: pages, pool */
    t0 = mzi_now_s();
    for (r = 0; r < reps; ++r) mzi_parallel_for(n, pack_grain(n), probe_range, &P);
    t0 = (mzi_now_s() - t0) / (reps > 0 ? reps : 1);
    free(hoA); free(buf); free(fmt); free(esz);
    return t0;
}
