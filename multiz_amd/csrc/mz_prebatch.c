/* mz_prebatch.c -- mz_preyama_batch(): N independent pre_yama() merges (reference mz_preyama.c:152-359, with mafBuild
 * :38-81 and mafScoreRange mz_scores.c:124-152) from block TEXT to block text -- what mz_multiz / mz_multic / mz_roast
 * run every merge through.
 *
 * Like mz_yama_batch() (mz_batch.c) this path is bounded by the PCIe link and the host's own memory traffic, not by
 * the kernels, and it is built the same way:
 *
 *   up     the text of the two slices as byte CLASSES, two per byte, every row a whole number of 32-byte lines
 *          (mz_pack_classes_stream; k_unnib expands a nibble to a canonical letter of its class on the device: the DP,
 *          rmColDash and mafScoreRange only distinguish A/a C/c G/g T/t, '-' and "other") -- no int32 band at all: the
 *          band is derived on the device (k_pre: the shared reference row walked base by base, then smooth());
 *   back   a 48-byte record per merge (status, M, N, OM, score), the bases per row, and per merged column one bit per
 *          row group (k_fin) -- the ROWS are put together here, from the caller's own text (mz_assemble_rows);
 *   pipe   a call is cut into chunks that go through the chunk pipeline of mz_flow.c on MZ_SETS rotating sets of buffers,
 *          chunk k on chunk stream k % nq, its link traffic as kernels of that stream (mzk_link_copy):
 *            cut        (the calling thread)  lays the chunk out; its packing -- classes into pinned memory -- is a loop
 *                                             posted to the pool's threads, among which the caller works;
 *            sender     (stage thread)        staging block -> device, k_unnib, k_pre, the first plan, its totals -> host;
 *            launcher 1 (stage thread)        waits for the totals, sizes the workspaces, issues DP / walk / emit; for
 *                                             chunks with two-stage merges k_mid and the second plan, else k_fin;
 *            launcher 2 (stage thread)        chunks with two-stage merges: waits for the second plan's totals, issues
 *                                             the second DP / walk / emit and k_fin;
 *            collector  (stage thread)        waits for the chunk's last kernel (the results -> host copy), takes ONE block
 *                                             for the chunk's rows; the assembling is a loop posted to the pool.
 *          No stage waits for a copy or a kernel another stage could work beside.  A call of one chunk runs inline.
 */
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <xmmintrin.h>

#include "mz_ctx.h"
#include "mz_pack.h"
#include "../../include/mz_scores.h"

typedef struct ppack { const mz_prejob *jobs; const int64_t *hoT1; uint8_t *hTxt; } ppack;
struct pasm;

typedef struct pchunk {
    mz_ctx *X;
    int set, n, index, any0, lane, wide, solo;   /* lane: which stream slot (X->qf / qd / qt[lane]); -1: the context's own stream (a call of one chunk); wide: four DP slots, the tail is the slot's own */
    const mz_prejob *jobs;
    mz_preout *outs;
    mz_dev_batch b, b2;
    mz_pre_batch q;
    mz_fin_batch f;
    ppack pc;
    const uint8_t *dNib;
    size_t txt, nrow_res;
    struct pasm *a;
    size_t *where;
    const int64_t *hoRow, *hoMask;         /* host copies (inside the set's pinned staging block) */
    size_t nrow, mask_bytes, in_bytes, res_bytes;
    int64_t cells;
    double t_pack0, t_pack1, t_packed, t_up, t_l1a, t_l1b, t_l2a, t_l2b, t_col0, t_col1, t_col2;
} pchunk;

static size_t row_stride(int cols) { return ((size_t)cols + 63) & ~(size_t)63; }
static size_t text_bytes(const mz_prejob *j) { return (size_t)j->K * row_stride(j->M_all) + (size_t)j->L1 * row_stride(j->N_all); }
static size_t mask_block(const mz_prejob *j)         /* bytes of a merge's mask block: the layout of k_fin (kernels/prepost.inc) */
{
    const size_t cap = j->v == 0 ? 2 * (size_t)j->M_all + j->N_all : (size_t)j->M_all + j->N_all, wc = (cap + 63) >> 6;
    return 8 * (j->v == 0 ? 3 * wc + (((size_t)j->M_all + 63) >> 6) + (((size_t)j->N_all + 63) >> 6) : 2 * wc + (((size_t)j->N_all + 63) >> 6));
}

static int pre_grain(int n)
{
    if (n <= 16) return n;
    const int g = n / (4 * mzi_pool_threads());
    return g < 1 ? 1 : g > 64 ? 64 : g;
}

static void pack_text(void *ctx, int lo, int hi)
{
    const ppack *q = (const ppack *)ctx;
    int p, k;
    for (p = lo; p < hi; ++p) {
        const mz_prejob *j = &q->jobs[p];
        const size_t sA = row_stride(j->M_all), sB = row_stride(j->N_all);
        uint8_t *d = q->hTxt + q->hoT1[p] / 2;
        if (p + 1 < hi) { _mm_prefetch((const char *)q->jobs[p + 1].rows1[0], _MM_HINT_T0); _mm_prefetch((const char *)q->jobs[p + 1].rows2[0], _MM_HINT_T0); }
        for (k = 0; k < j->K; ++k, d += sA / 2) mz_pack_classes_stream((const uint8_t *)j->rows1[k], (size_t)j->M_all, d, sA / 2);
        for (k = 0; k < j->L1; ++k, d += sB / 2) mz_pack_classes_stream((const uint8_t *)j->rows2[k], (size_t)j->N_all, d, sB / 2);
    }
    _mm_sfence();
}

/* the chunk's streams (mz_ctx.h): front (staging block -> device, k_pre, plan), DP, tail (walk, emit, k_mid + second plan, k_fin, results -> host) */
static hipStream_t pchunk_front(const pchunk *c) { return c->lane < 0 ? c->X->stream : c->X->qf[c->index % c->X->nf]; }
static hipStream_t pchunk_dp(const pchunk *c) { return c->lane < 0 ? c->X->stream : c->X->qd[c->lane]; }
static hipStream_t pchunk_tail(const pchunk *c) { return c->lane < 0 ? c->X->stream : c->wide ? c->X->qt[c->lane] : c->X->qt[c->index % c->X->nt]; }
static int g_ptiming = -1;
#define PSTAMP(X, set, k, st) do { if (g_ptiming >= 2 && (X)->ptime_ready) HIPCK(hipEventRecord((X)->ptime[set][k], st)); } while (0)
#define PD(i) (&X->pd[set][i])
#define PH(i) (&X->ph[set][i])

/* the calling thread: chunk `index` = the n merges at `jobs`, laid out in buffer set `set`; its packing as a loop (*pack) */
static int pchunk_cut(mz_ctx *X, pchunk *c, int index, int set, int lane, int n, const mz_prejob *jobs, mz_preout *outs, mz_ajob *pack)
{
    size_t txt = 0, szA = 0, szB = 0, szA2 = 0, nband = 0, nscr = 0, nrow = 0, nmask = 0, nprep = 0, nprep2 = 0, hdr, in_bytes, lds16 = 0, lds32 = 0, res_bytes;
    int wmax = 1, cmax = 0;
    int64_t *hT1, *hoA, *hoB, *hoBand, *hoScr, *hoRow, *hoA2, *hoMask;
    int32_t *hK, *hL, *hMa, *hNa, *hRad, *hV;
    uint8_t *hTxt;
    const uint8_t *dNib;
    char *h, *d;
    int p, any0 = 0;

    c->t_pack0 = mzi_now_s();
    c->X = X; c->set = set; c->index = index; c->lane = lane < 0 ? -1 : (lane & 0xff); c->wide = lane >= 0 && ((lane >> 8) & 1); c->solo = lane >= 0 && ((lane >> 9) & 1); c->n = n; c->jobs = jobs; c->outs = outs; c->cells = 0;
    for (p = 0; p < n; ++p) {
        const mz_prejob *j = &jobs[p];
        txt += text_bytes(j);
        szA += (size_t)j->K * j->M_all; szB += (size_t)(j->L1 - 1) * j->N_all; szA2 += (size_t)j->M_all + 8;
        nband += (size_t)j->M_all + 1; nscr += 8 * ((size_t)j->M_all + 2) + 6 * ((size_t)j->N_all + 2);
        nrow += (size_t)j->K + j->L1 - 1; nmask += mask_block(j);
        nprep += MZ_PREP_BOUND(j->N_all);                                       /* (the first yama(): at most N_all columns) */
        if (j->v == 0) { any0 = 1; nprep2 += MZ_PREP_BOUND((size_t)j->M_all + j->N_all); }    /* (the second: its B is the first one's result) */
        {   /* what k_pre would need of LDS for this pair (int16 / int32 scratch), the widest merged block, the longest slice */
            const int W = j->K + j->L1 - 1;
            const size_t a = MZ_PRE_LDS(text_bytes(j), j->M_all, j->N_all, W, 2), b32 = MZ_PRE_LDS(text_bytes(j), j->M_all, j->N_all, W, 4);
            if (a > lds16) lds16 = a;
            if (b32 > lds32) lds32 = b32;
            if (W > wmax) wmax = W;
            if (j->M_all > cmax) cmax = j->M_all;
            if (j->N_all > cmax) cmax = j->N_all;
        }
    }
    /* pinned staging: K L Ma Na rad v (int32 x n), offT1 offA offB offBand offScr offRow offA2 offMask (int64 x n), class nibbles */
    hdr = 6 * mzi_al256(4 * (size_t)n) + 8 * mzi_al256(8 * (size_t)n);
    in_bytes = hdr + mzi_al256(txt / 2);
    /* the text on the device: kept in LDS as nibbles by k_pre when the chunk's largest pair fits (MZ_PRE_LDS=0: never), else
     * expanded to a byte per class in HBM first (k_unnib) */
    {
        static int lds_on = -1;
        if (lds_on < 0) { const char *e = getenv("MZ_PRE_LDS"); lds_on = !(e && e[0] == '0'); }
        c->q.lds16 = cmax < 32768;
        c->q.lds_bytes = !lds_on ? 0 : c->q.lds16 ? (lds16 <= 65536 ? (int)lds16 : 0) : (lds32 <= 65536 ? (int)lds32 : 0);
        if (c->q.lds16 && !c->q.lds_bytes && lds_on && lds32 <= 65536) { c->q.lds16 = 0; c->q.lds_bytes = (int)lds32; }
    }
    res_bytes = mzi_al256(64 + mzi_al256(sizeof(mz_pre_rec) * (size_t)n) + mzi_al256(4 * nrow) + mzi_al256(nmask));
    if (mzi_host_reserve(PH(MZ_PH_IN), in_bytes) || mzi_dev_reserve(PD(MZ_PD_IN), in_bytes) || (!c->q.lds_bytes && mzi_dev_reserve(PD(MZ_PD_TXT), txt + 256)) ||
        mzi_dev_reserve(PD(MZ_PD_RES), res_bytes) || mzi_host_reserve(PH(MZ_PH_RES), res_bytes) ||
        mzi_dev_reserve(PD(MZ_PD_COLS), mzi_al256(szA) + mzi_al256(szB) + 256) || mzi_dev_reserve(PD(MZ_PD_BAND), 2 * mzi_al256(4 * nband)) ||
        mzi_dev_reserve(PD(MZ_PD_SCR), 4 * nscr + 256)) return -1;
    h = (char *)PH(MZ_PH_IN)->p; d = (char *)PD(MZ_PD_IN)->p;
    {
        const int keep16 = c->q.lds16, keepb = c->q.lds_bytes;            /* (decided above; the structures are cleared next) */
        memset(&c->b, 0, sizeof c->b); memset(&c->b2, 0, sizeof c->b2); memset(&c->q, 0, sizeof c->q); memset(&c->f, 0, sizeof c->f);
        c->q.lds16 = keep16; c->q.lds_bytes = keepb;
    }
    c->b.n = c->b2.n = c->q.n = n;
    c->b.dp_hint = c->b2.dp_hint = c->lane >= 0 ? MZ_DP_HELPERS_FIRST | (c->solo ? MZ_DP_SOLO : 0) : 0;      /* (beside other chunks' DPs: include/mz_amd.h) */
#define SL(hptr, type, dptr, bytes) do { hptr = (type *)h; dptr = (const type *)d; h += mzi_al256(bytes); d += mzi_al256(bytes); } while (0)
    SL(hK, int32_t, c->q.K, 4 * (size_t)n); SL(hL, int32_t, c->q.L, 4 * (size_t)n); SL(hMa, int32_t, c->q.Ma, 4 * (size_t)n);
    SL(hNa, int32_t, c->q.Na, 4 * (size_t)n); SL(hRad, int32_t, c->q.rad, 4 * (size_t)n); SL(hV, int32_t, c->q.v, 4 * (size_t)n);
    SL(hT1, int64_t, c->q.offT1, 8 * (size_t)n);
    SL(hoA, int64_t, c->b.offA, 8 * (size_t)n); SL(hoB, int64_t, c->b.offB, 8 * (size_t)n); SL(hoBand, int64_t, c->b.offBand, 8 * (size_t)n);
    SL(hoScr, int64_t, c->q.offScr, 8 * (size_t)n); SL(hoRow, int64_t, c->f.offRow, 8 * (size_t)n); SL(hoA2, int64_t, c->b2.offA, 8 * (size_t)n);
    SL(hoMask, int64_t, c->f.offMask, 8 * (size_t)n);
    SL(hTxt, uint8_t, dNib, txt / 2);
#undef SL
    {
        size_t ot = 0, oa = 0, ob = 0, od = 0, os = 0, orow = 0, oa2 = 0, om = 0;
        for (p = 0; p < n; ++p) {
            const mz_prejob *j = &jobs[p];
            hK[p] = j->K; hL[p] = j->L1 - 1; hMa[p] = j->M_all; hNa[p] = j->N_all; hRad[p] = j->radius; hV[p] = j->v;
            hT1[p] = (int64_t)ot; ot += text_bytes(j);
            hoA[p] = (int64_t)oa; oa += (size_t)j->K * j->M_all;
            hoB[p] = (int64_t)ob; ob += (size_t)(j->L1 - 1) * j->N_all;     /* (upper bound: dash columns go on the device) */
            hoBand[p] = (int64_t)od; od += (size_t)j->M_all + 1;            /* (both stages: the second job has at most M_all rows) */
            hoScr[p] = (int64_t)os; os += 8 * ((size_t)j->M_all + 2) + 6 * ((size_t)j->N_all + 2);
            hoRow[p] = (int64_t)orow; orow += (size_t)j->K + j->L1 - 1;
            hoA2[p] = (int64_t)oa2; oa2 += (size_t)j->M_all + 8;            /* the first block's top row without its dashes */
            hoMask[p] = (int64_t)om; om += mask_block(j);
        }
    }
    c->hoRow = hoRow; c->hoMask = hoMask; c->nrow = nrow; c->mask_bytes = nmask; c->in_bytes = in_bytes; c->any0 = any0;
    c->dNib = dNib; c->txt = txt;
    c->q.nib = dNib; c->q.stride64 = 1;
    if (!c->q.lds_bytes) c->q.txt = (const uint8_t *)PD(MZ_PD_TXT)->p;
    {   /* the results' device image: 64-byte header, a record per merge, bases per row, the mask blocks (k_pre fills in the
         * counts and rmColDash's verdicts, k_fin the rest) */
        char *dres = (char *)PD(MZ_PD_RES)->p;
        c->f.any0 = any0;
        c->f.recs = (mz_pre_rec *)(dres + 64);
        c->f.size = (int32_t *)(dres + 64 + mzi_al256(sizeof(mz_pre_rec) * (size_t)n));
        c->f.masks = (uint8_t *)c->f.size + mzi_al256(4 * nrow);
        c->f.cols = MZ_FIN_COLS(wmax); c->f.lds_bytes = MZ_FIN_LDS(wmax);
        c->res_bytes = res_bytes;
    }
    c->b.poolA = (const uint8_t *)PD(MZ_PD_COLS)->p; c->b.poolB = c->b.poolA + mzi_al256(szA);
    c->b.poolLB = (const int32_t *)PD(MZ_PD_BAND)->p; c->b.poolRB = (const int32_t *)((char *)PD(MZ_PD_BAND)->p + mzi_al256(4 * nband));
    c->q.scr = (int32_t *)PD(MZ_PD_SCR)->p;
    /* K L M N of the two device batches, offB of the second, the NULL flags: one more device block */
    if (mzi_dev_reserve(PD(MZ_PD_META), 9 * mzi_al256(4 * (size_t)n) + mzi_al256(8 * (size_t)n))) return -1;
    {
        char *e = (char *)PD(MZ_PD_META)->p;
#define TK(field) do { field = (const int32_t *)e; e += mzi_al256(4 * (size_t)n); } while (0)
        TK(c->b.K); TK(c->b.L); TK(c->b.M); TK(c->b.N); TK(c->b2.K); TK(c->b2.L); TK(c->b2.M); TK(c->b2.N);
#undef TK
        c->q.nullres = (int32_t *)e; e += mzi_al256(4 * (size_t)n);
        c->b2.offB = (const int64_t *)e;
    }
    if (any0) {       /* the second stage's A (the top rows) and band pools; its B is the first stage's merged columns where they lie */
        if (mzi_dev_reserve(PD(MZ_PD_A2), szA2 + 256) || mzi_dev_reserve(PD(MZ_PD_BAND2), 2 * mzi_al256(4 * nband))) return -1;
        c->b2.poolA = (const uint8_t *)PD(MZ_PD_A2)->p;
        c->b2.poolLB = (const int32_t *)PD(MZ_PD_BAND2)->p; c->b2.poolRB = (const int32_t *)((char *)PD(MZ_PD_BAND2)->p + mzi_al256(4 * nband));
        c->b2.offBand = c->b.offBand;
    }
    if (mzi_dev_reserve(&X->d_plan[set], mz_dev_plan_bytes(n)) || mzi_host_reserve(&X->h_tot[set], 16 * sizeof(int64_t)) ||
        mzi_dev_reserve(&X->d_prep[set], 4 * nprep + 256) || (any0 && mzi_dev_reserve(PD(MZ_PD_PREP2), 4 * nprep2 + 256))) return -1;
    mz_dev_carve(&c->b, X->d_plan[set].p);
    c->b.capTb = c->b.capScript = c->b.capOut = INT64_MAX;
    /* (the prep records' size is bounded by the slices' columns: MZ_PREP_BOUND -- they are made right behind the plan) */
    c->b.prep = (uint32_t *)X->d_prep[set].p; c->b.capPrep = (int64_t)(X->d_prep[set].cap / 4);
    c->pc.jobs = jobs; c->pc.hoT1 = hT1; c->pc.hTxt = hTxt;
    pack->fn = pack_text; pack->ctx = &c->pc; pack->n = n; pack->grain = pre_grain(n); pack->hedge = 1;
    c->t_pack1 = mzi_now_s();
    return 0;
}

/* stage 1: the chunk is packed -- staging block -> device (a kernel on the chunk's stream), k_unnib where the text does not fit
 * the LDS, k_pre, the first plan, its totals -> host */
static int pchunk_send(pchunk *c)
{
    mz_ctx *X = c->X;
    const int set = c->set;
    hipStream_t st = pchunk_front(c);
    c->t_packed = mzi_now_s();
    {
        hipStream_t sc = c->lane < 0 ? st : X->qc;           /* the link copy: a stream of its own (mz_flow.c: which pipe) */
        PSTAMP(X, set, 0, sc);
        if (mzk_link_copy(PD(MZ_PD_IN)->p, PH(MZ_PH_IN)->p, mzi_al256(c->in_bytes), sc)) return mzi_set_err("%s", mzk_last_error());
        PSTAMP(X, set, 1, sc);
        if (sc != st) { HIPCK(hipEventRecord(X->bcopy[set], sc)); HIPCK(hipStreamWaitEvent(st, X->bcopy[set], 0)); }
    }
    if (!c->q.lds_bytes) {
        if (mzk_unnib(c->dNib, PD(MZ_PD_TXT)->p, (long long)c->txt, st)) return mzi_set_err("%s", mzk_last_error());
        HIPCK(hipMemsetAsync(c->f.size, 0, 4 * c->nrow, st));      /* (the LDS-free k_pre adds the row counts up in place) */
    }
    if (mzk_pre(&c->q, &c->f, &c->b, st)) return mzi_set_err("%s", mzk_last_error());
    PSTAMP(X, set, 2, st);
    if (mzk_plan(&c->b, st) || mzk_link_copy(X->h_tot[set].p, c->b.totals, 16 * sizeof(int64_t), st)) return mzi_set_err("%s", mzk_last_error());
    HIPCK(hipEventRecord(X->bplan[set], st));
    if (mzk_prep(&c->b, st)) return mzi_set_err("%s", mzk_last_error());
    HIPCK(hipEventRecord(X->bprep[set], st));
    PSTAMP(X, set, 3, st);
    c->t_up = mzi_now_s();
    return 0;
}

/* size one stage's workspaces from its plan's totals and issue DP / walk / emit (the prep records were made behind the plan: bprep) */
static int run_stage(mz_ctx *X, int set, int stamp, hipStream_t sd, hipStream_t st, int wait_prep, const mz_dp_lanes *lanes, mz_dev_batch *b, int n, const int64_t *totals, gbuf *tb, gbuf *script, gbuf *out)
{
    if (mzi_dev_reserve(tb, 4 * (size_t)totals[0] + 256) || mzi_dev_reserve(script, (size_t)totals[1] + 256) ||
        mzi_dev_reserve(out, (size_t)totals[2] + 256)) return -1;
    b->tbw = (uint32_t *)tb->p; b->script = (uint8_t *)script->p; b->out = (uint8_t *)out->p;
    b->capTb = (int64_t)(tb->cap / 4); b->capScript = (int64_t)script->cap; b->capOut = (int64_t)out->cap;
    b->walk_hint = mz_walk_choice(n, totals);
    b->dp_hint = mz_dp_hint(n, totals) | (b->dp_hint & MZ_DP_REQUESTS); b->dp_grid = mz_dp_grid(n, totals); b->dp_rows = mz_dp_rows(n, totals); b->hint_gen = g_hint_gen;
    if (lanes && !X->lanes_made && mz_dp_kinds(b->dp_hint) > 1 && mzi_flow_lanes(X)) return -1;      /* several kinds of pairs: the DP streams' lanes */
    /* the DP on the slot's DP stream (whatever else it reads is through: the host has seen this plan's totals), the rest behind its event */
    /* (the prep records are waited for HERE and the DP's "done" event rides on its own dispatch packet where it can: every packet on the DP
     *  stream starts 60-200 us late while its pipe-mate, the tail of the chunk before, has a kernel running -- mz_batch.c, chunk_launch) */
    if (wait_prep) HIPCK(hipEventSynchronize(X->bprep[set]));
    {
        /* (the two launch stages run on threads of their own and chunks k and k + 2 share a DP stream's lanes: the fork / launch / join
         *  sequence of one at a time -- what a wait refers to is fixed when it is enqueued) */
        static pthread_mutex_t launch_mu = PTHREAD_MUTEX_INITIALIZER;
        int rc, rides = 0;
        pthread_mutex_lock(&launch_mu);
        rc = mzk_dp_range_ev(b, 0, n, sd, lanes, st != sd ? (void *)X->bdp[set] : NULL, &rides);
        pthread_mutex_unlock(&launch_mu);
        if (rc) return mzi_set_err("%s", mzk_last_error());
        if (stamp >= 0) PSTAMP(X, set, stamp, sd);
        if (st != sd) { if (!rides) HIPCK(hipEventRecord(X->bdp[set], sd)); HIPCK(hipStreamWaitEvent(st, X->bdp[set], 0)); }
    }
    if (mzk_walk(b, st, 1) || mzk_emit(b, st)) return mzi_set_err("%s", mzk_last_error());
    if (stamp >= 0) PSTAMP(X, set, stamp + 1, st);
    return 0;
}

/* k_fin (the results' device image was laid out at upload time: k_pre has written part of it) */
static int pchunk_finish(pchunk *c)
{
    mz_ctx *X = c->X;
    const int set = c->set;
    hipStream_t st = pchunk_tail(c);
    PSTAMP(X, set, 6, st);
    if (mzk_fin(&c->q, &c->f, &c->b, &c->b2, st)) return mzi_set_err("%s", mzk_last_error());
    PSTAMP(X, set, 7, st);
    /* the results go home as the chunk's last kernel (round 4 had the collector issue a copy once k_fin was done: a copy engine
     * takes its commands in order across all streams) */
    if (mzk_link_copy(PH(MZ_PH_RES)->p, PD(MZ_PD_RES)->p, c->res_bytes, st)) return mzi_set_err("%s", mzk_last_error());
    HIPCK(hipEventRecord(X->bdone[set], st));
    return 0;
}

/* stage 2: the first yama() of every merge */
static int pchunk_launch1(pchunk *c)
{
    mz_ctx *X = c->X;
    const int set = c->set, n = c->n;
    hipStream_t st = pchunk_tail(c);
    const mz_dp_lanes *lanes = c->lane < 0 ? NULL : &X->qlane[c->lane];
    c->t_l1a = mzi_now_s();
    HIPCK(hipEventSynchronize(X->bplan[set]));
    if (run_stage(X, set, 4, pchunk_dp(c), st, pchunk_dp(c) != pchunk_front(c), lanes, &c->b, n, (const int64_t *)X->h_tot[set].p, &X->d_tb[set], &X->d_script[set], PD(MZ_PD_OUT1))) return -1;
    if (c->any0) {
        /* the second yama() job of the two-stage merges, derived where the first one's result lies (k_mid) */
        c->b2.poolB = c->b.out;
        if (mzk_mid(&c->q, &c->b, &c->b2, st)) return mzi_set_err("%s", mzk_last_error());
        if (mzi_dev_reserve(PD(MZ_PD_PLAN2), mz_dev_plan_bytes(n)) || mzi_host_reserve(PH(MZ_PH_TOT2), 16 * sizeof(int64_t))) return -1;
        mz_dev_carve(&c->b2, PD(MZ_PD_PLAN2)->p);
        c->b2.capTb = c->b2.capScript = c->b2.capOut = INT64_MAX;
        c->b2.prep = (uint32_t *)PD(MZ_PD_PREP2)->p; c->b2.capPrep = (int64_t)(PD(MZ_PD_PREP2)->cap / 4);
        if (mzk_plan(&c->b2, st) || mzk_link_copy(PH(MZ_PH_TOT2)->p, c->b2.totals, 16 * sizeof(int64_t), st)) return mzi_set_err("%s", mzk_last_error());
        HIPCK(hipEventRecord(X->pplan2[set], st));
        if (mzk_prep(&c->b2, st)) return mzi_set_err("%s", mzk_last_error());
        HIPCK(hipEventRecord(X->bprep[set], st));            /* (the first stage's DP has long taken its records) */
    } else if (pchunk_finish(c)) return -1;
    c->t_l1b = mzi_now_s();
    return 0;
}

/* stage 3: the second yama() of the two-stage merges (chunks without any pass through) */
static int pchunk_launch2(pchunk *c)
{
    mz_ctx *X = c->X;
    const int set = c->set;
    c->t_l2a = c->t_l2b = mzi_now_s();
    if (!c->any0) return 0;
    HIPCK(hipEventSynchronize(X->pplan2[set]));
    if (run_stage(X, set, -1, pchunk_dp(c), pchunk_tail(c), pchunk_dp(c) != pchunk_tail(c), c->lane < 0 ? NULL : &X->qlane[c->lane], &c->b2, c->n, (const int64_t *)PH(MZ_PH_TOT2)->p, PD(MZ_PD_TB2), PD(MZ_PD_SCRIPT2), PD(MZ_PD_OUT2))) return -1;
    if (pchunk_finish(c)) return -1;
    c->t_l2b = mzi_now_s();
    return 0;
}

/* stage 4: results -> rows */
typedef struct pasm {
    const mz_prejob *jobs;
    mz_preout *outs;
    const mz_pre_rec *rec;
    const int32_t *size;
    const uint8_t *masks;
    const int64_t *hoRow, *hoMask;
    const size_t *where;
    uint8_t *block;
    int failed, oom;
    int64_t cells;
} pasm;

#define PASM_ROWS 32
static void assemble_merges(void *ctx, int lo, int hi)
{
    pasm *q = (pasm *)ctx;
    int p;
    for (p = lo; p < hi; ++p) {                          /* (a piece may be run twice -- mz_pool.c -- so it adds nothing up: pchunk_done() does) */
        const mz_prejob *j = &q->jobs[p];
        const mz_pre_rec *r = &q->rec[p];
        mz_preout *o = &q->outs[p];
        const int W = j->K + j->L1 - 1, two = j->v == 0;
        /* (every field gets its final value in one store, and o->block is not this loop's: pchunk_done() hangs the chunk's block on its
         *  first merge when the loop is complete, and a piece that is run a second time may still be at work then -- mz_pool.c) */
        o->null_result = r->nullres; o->status = r->status; o->badrow = r->badrow; o->stage = r->stage; o->M = r->M; o->N = r->N;
        if (r->nullres || r->status != MZ_OK) { o->OM = 0; o->score = 0; o->size = NULL; o->rows = NULL; continue; }
        o->OM = r->om;
        o->score = (double)r->score;
        {
            const size_t nb = (size_t)W * (size_t)r->om, pad = (nb + 7) & ~(size_t)7;
            const size_t cap = two ? 2 * (size_t)j->M_all + j->N_all : (size_t)j->M_all + j->N_all, wc = (cap + 63) >> 6;
            const uint64_t *mk = (const uint64_t *)(q->masks + q->hoMask[p]);
            const uint64_t *opsA = mk, *opsB = mk + wc, *opsT = mk + 2 * wc, *keepA = mk + 3 * wc,
                           *keepB = two ? keepA + (((size_t)j->M_all + 63) >> 6) : mk + 2 * wc;
            const int sqA = two && r->M < j->M_all, sqB = r->N < j->N_all;      /* rmColDash dropped columns of that slice */
            mz_rowspec small[PASM_ROWS], *rows = W <= PASM_ROWS ? small : (mz_rowspec *)malloc((size_t)W * sizeof *rows);
            const size_t longest = (size_t)(j->M_all > j->N_all ? j->M_all : j->N_all);
            uint8_t tmp_small[4096 + 16], *tmp = NULL;
            int k;
            if (two || sqB) tmp = longest + 16 <= sizeof tmp_small ? tmp_small : (uint8_t *)malloc(longest + 16);
            if (!rows || ((two || sqB) && !tmp)) { if (rows != small) free(rows); q->oom = 1; o->status = MZ_E_DEVICE; continue; }
            for (k = 0; k < j->K; ++k) {
                mz_rowspec *s = &rows[k];
                s->src = (const uint8_t *)j->rows1[k]; s->n = j->M_all; s->tmp = tmp;
                if (two && k == 0) { s->squeeze = 2; s->keep = NULL; s->ops = opsT; }      /* the first block's top row: its bases */
                else { s->squeeze = sqA; s->keep = keepA; s->ops = opsA; }
            }
            for (k = 1; k < j->L1; ++k) {
                mz_rowspec *s = &rows[j->K + k - 1];
                s->src = (const uint8_t *)j->rows2[k]; s->n = j->N_all; s->tmp = tmp;
                s->squeeze = sqB; s->keep = keepB; s->ops = opsB;
            }
            o->rows = q->block + q->where[p];
            if (p + 1 < hi) { _mm_prefetch((const char *)q->jobs[p + 1].rows1[0], _MM_HINT_T0); _mm_prefetch((const char *)q->jobs[p + 1].rows2[0], _MM_HINT_T0); }
            mz_assemble_rows(W, rows, r->om, o->rows);
            memcpy(o->rows + pad, q->size + q->hoRow[p], 4 * (size_t)W);
            o->size = (const int *)(o->rows + pad);
            if (rows != small) free(rows);
            if (tmp && tmp != tmp_small) free(tmp);
        }
    }
    _mm_sfence();
}

/* stage 4: wait for the chunk's last kernel (the results are in host memory then), take ONE block for the chunk's rows and base
 * counts; the assembling itself is the loop *post */
static int pchunk_collect(pchunk *c, mz_ajob *post)
{
    mz_ctx *X = c->X;
    const int n = c->n, set = c->set;
    const char *r = (const char *)PH(MZ_PH_RES)->p;
    const mz_pre_rec *rec = (const mz_pre_rec *)(r + 64);
    size_t *where, total = 0;
    uint8_t *block = NULL;
    pasm *a;
    int p;

    c->t_col0 = mzi_now_s();
    HIPCK(hipEventSynchronize(X->bdone[set]));
    c->t_col1 = mzi_now_s();
    free(c->where); c->where = NULL;
    if (!c->a && !(c->a = (pasm *)malloc(sizeof *c->a))) return mzi_set_err("out of memory");
    a = c->a;
    where = (size_t *)malloc(((size_t)n + 1) * sizeof *where);
    if (!where) return mzi_set_err("out of memory");
    /* ONE allocation for the chunk's rows and base counts; outs[first].block owns it (mz_free_preouts).  Every merge starts on
     * a 64-byte line of its own (no line shared between two host threads; whole-line streaming stores) */
    for (p = 0; p < n; ++p) {
        const int W = c->jobs[p].K + c->jobs[p].L1 - 1;
        where[p] = total;
        if (rec[p].nullres || rec[p].status != MZ_OK) continue;
        total += ((((size_t)W * (size_t)rec[p].om + 7) & ~(size_t)7) + 4 * (size_t)W + 63) & ~(size_t)63;
    }
    if (total) {
        size_t mis;
        block = (uint8_t *)mzi_block_get(total + 128);
        if (!block) { free(where); return mzi_set_err("out of memory for the merged rows (%zu bytes)", total); }
        mis = (size_t)(-(intptr_t)block & 63);
        for (p = 0; p < n; ++p) where[p] += mis;
    }
    c->where = where;
    a->jobs = c->jobs; a->outs = c->outs; a->rec = rec;
    a->size = (const int32_t *)(r + 64 + mzi_al256(sizeof(mz_pre_rec) * (size_t)n));
    a->masks = (const uint8_t *)a->size + mzi_al256(4 * c->nrow);
    a->hoRow = c->hoRow; a->hoMask = c->hoMask; a->where = where; a->block = block; a->failed = 0; a->oom = 0; a->cells = 0;
    post->fn = assemble_merges; post->ctx = a; post->n = n; post->grain = pre_grain(n); post->hedge = 1;
    return 0;
}

static int pchunk_done(pchunk *c)
{
    pasm *a = c->a;
    int p;
    a->failed = 0; a->cells = 0;
    for (p = 0; p < c->n; ++p) { a->failed += c->outs[p].null_result || c->outs[p].status != MZ_OK; a->cells += a->rec[p].cells; }
    c->cells = a->cells;
    c->outs[0].block = a->block;
    c->t_col2 = mzi_now_s();
    if (a->oom) return mzi_set_err("out of memory for the merged rows");
    return a->failed;
}

/* ------------------------------------------------------------------------------------------------ the pipeline (mz_flow.c) */

typedef struct ppipe {
    mz_ctx *X;
    int n, up, max_pairs, threaded, slots;   /* slots: DP streams of this call (2; 4 for a call of few long merges: mz_batch.c) */
    size_t max_bytes;
    const mz_prejob *jobs;
    mz_preout *outs;
    pchunk ck[MZ_SETS];
    pthread_mutex_t mu;                    /* the sums, the report lines */
    int64_t cells, bytes_up, bytes_down;
    double t0;
    hipEvent_t ev0;                        /* MZ_TIMING=2: recorded on the context's stream when the call starts */
} ppipe;

static void pchunk_report(const ppipe *P, const pchunk *c)
{
    float g[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    int k;
    if (g_ptiming < 2) return;
    /* GPU time stamps of the chunk's stream against the call's start (ms): upload begins / ends, k_pre done, planned, first DP done,
     * first walk + emit done, k_fin begins / ends */
    if (c->X->ptime_ready) { hipSetDevice(c->X->device); for (k = 0; k < 8; ++k) hipEventElapsedTime(&g[k], P->ev0, c->X->ptime[c->set][k]); }
    fprintf(stderr, "{\"mz_preyama_batch_chunk\": %d, \"merges\": %d, \"stream\": %d, \"two_stage\": %d, \"cells\": %lld, \"bytes_up\": %zu, \"bytes_down\": %zu, "
                    "\"host_ms\": {\"cut\": [%.3f, %.3f], \"packed\": %.3f, \"sent\": %.3f, \"launch1\": [%.3f, %.3f], \"launch2\": [%.3f, %.3f], \"result_wait\": [%.3f, %.3f], \"assembled\": %.3f}, "
                    "\"gpu_ms\": {\"h2d\": [%.3f, %.3f], \"pre_done\": %.3f, \"planned\": %.3f, \"dp_done\": %.3f, \"emit_done\": %.3f, \"fin\": [%.3f, %.3f]}}\n",
            c->index, c->n, c->lane, c->any0, (long long)c->cells, c->in_bytes, c->res_bytes,
            1e3 * (c->t_pack0 - P->t0), 1e3 * (c->t_pack1 - P->t0), 1e3 * (c->t_packed - P->t0), 1e3 * (c->t_up - P->t0), 1e3 * (c->t_l1a - P->t0), 1e3 * (c->t_l1b - P->t0),
            1e3 * (c->t_l2a - P->t0), 1e3 * (c->t_l2b - P->t0), 1e3 * (c->t_col0 - P->t0), 1e3 * (c->t_col1 - P->t0), 1e3 * (c->t_col2 - P->t0),
            g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7]);
}

/* pieces a GPU's share of a call is cut into (MZ_CHUNKS overrides): chunks of about 5 000 merges, three to sixteen of them, as
 * mz_yama_batch() (mz_batch.c: chunk_parts) -- of about 10 000 when most of the merges are two-stage ones: a chunk of those goes through
 * two plans and two sets of DP / walk / emit, every stage with its wait for the stage before */
static int pre_parts(int n, int two_stage)
{
    static int v = -1;
    int parts;
    if (v < 0) { const char *e = getenv("MZ_CHUNKS"); v = e && atoi(e) > 0 ? atoi(e) : 0; }
    if (v) return v;
    parts = 2 * two_stage >= n ? (n + 5000) / 10000 : (n + 2500) / 5000;
    parts = parts < 3 ? 3 : parts > 16 ? 16 : parts;
    if (n <= 4096 && (parts & 1)) ++parts;                /* (few merges, long ones as likely as not: an even number of chunks -- mz_batch.c, chunk_parts) */
    return parts;
}

#define PRE_MIN_CHUNK 1024
static int next_pchunk(const mz_prejob *jobs, int n, int first, int limit, size_t max_bytes, int min_pairs)
{
    size_t bytes = 0;
    int m = 0;
    while (first + m < n && m < limit && (bytes < max_bytes || (m < min_pairs && bytes < ((size_t)1 << 30)))) {
        bytes += text_bytes(&jobs[first + m]);
        ++m;
    }
    return m;
}

static int p_cut(void *self, int k, int set, mz_ajob *pack)
{
    ppipe *P = (ppipe *)self;
    /* the first chunks are a quarter and a half of the size: the GPU starts that much earlier */
    const int ramp = P->threaded && P->slots <= 2 && k < 2 && (P->max_pairs >= 2048 || P->max_bytes >= ((size_t)32 << 20)) ? 2 - k : 0;      /* (four DP slots: equal chunks, all side by side) */
    const int limit = P->max_pairs >> ramp < PRE_MIN_CHUNK / 2 ? PRE_MIN_CHUNK / 2 : P->max_pairs >> ramp;
    int m;
    if (P->up >= P->n) return 0;
    m = P->slots > 2 ? next_pchunk(P->jobs, P->n, P->up, (P->n + P->slots - 1) / P->slots, (size_t)1 << 30, 0)       /* (four DP slots: four chunks, side by side) */
                     : next_pchunk(P->jobs, P->n, P->up, limit, P->max_bytes >> ramp, PRE_MIN_CHUNK);
    if (pchunk_cut(P->X, &P->ck[set], k, set, P->threaded ? (k % P->slots) | (P->slots > 2 ? 0x100 | (P->n <= 1024 ? 0x200 : 0) : 0) : -1, m, P->jobs + P->up, P->outs + P->up, pack) < 0) return -1;
    P->up += m;
    return 1;
}
static int p_send(void *self, int k, int set, mz_ajob *post) { (void)k; (void)post; return pchunk_send(&((ppipe *)self)->ck[set]); }
static int p_l1(void *self, int k, int set, mz_ajob *post) { (void)k; (void)post; return pchunk_launch1(&((ppipe *)self)->ck[set]); }
static int p_l2(void *self, int k, int set, mz_ajob *post) { (void)k; (void)post; return pchunk_launch2(&((ppipe *)self)->ck[set]); }
static int p_collect(void *self, int k, int set, mz_ajob *post) { (void)k; return pchunk_collect(&((ppipe *)self)->ck[set], post); }
static int p_finish(void *self, int k, int set)
{
    ppipe *P = (ppipe *)self;
    pchunk *c = &P->ck[set];
    const int failed = pchunk_done(c);
    (void)k;
    pthread_mutex_lock(&P->mu);
    pchunk_report(P, c);
    P->cells += c->cells; P->bytes_up += (int64_t)c->in_bytes; P->bytes_down += (int64_t)c->res_bytes;
    pthread_mutex_unlock(&P->mu);
    return failed;
}

/* A call's share of one GPU, any size: MZ_SETS chunks at most are in flight.  stats: cells, bytes up, bytes down. */
int mzi_pre_on_ctx(mz_ctx *X, int n, const mz_prejob *jobs, mz_preout *outs, int64_t stats[3])
{
    ppipe *P;
    mz_flow *F;
    size_t max_bytes = 0, total_bytes = 0;
    int rc, s, max_pairs, two_stage = 0, parts;
    static int env_pairs = -1;

    if (env_pairs < 0) { const char *e = getenv("MZ_CHUNK_PAIRS"); env_pairs = e && atoi(e) > 0 ? atoi(e) : 0; }
    if (hipSetDevice(X->device) != hipSuccess) return mzi_set_err("hipSetDevice(%d) failed", X->device);
    P = (ppipe *)calloc(1, sizeof *P);
    F = (mz_flow *)calloc(1, sizeof *F);
    if (!P || !F) { free(P); free(F); return mzi_set_err("out of memory"); }
    P->X = X; P->n = n; P->jobs = jobs; P->outs = outs; P->t0 = mzi_now_s();
    pthread_mutex_init(&P->mu, NULL);
    if (g_ptiming >= 2) {
        if (!X->ptime_ready) {
            int a, e, ok = 1;
            for (a = 0; a < MZ_SETS && ok; ++a) for (e = 0; e < 8 && ok; ++e) ok = hipEventCreate(&X->ptime[a][e]) == hipSuccess;
            X->ptime_ready = ok;
        }
        if (X->ptime_ready && hipEventCreate(&P->ev0) == hipSuccess) hipEventRecord(P->ev0, X->stream);
        else X->ptime_ready = 0;
    }
    for (s = 0; s < n; ++s) { max_bytes += text_bytes(&jobs[s]); two_stage += jobs[s].v == 0; }
    total_bytes = max_bytes;
    parts = pre_parts(n, two_stage);
    max_bytes = max_bytes / (size_t)parts + 1;
    if (max_bytes < ((size_t)8 << 20)) max_bytes = (size_t)8 << 20;
    if (max_bytes > ((size_t)1 << 30)) max_bytes = (size_t)1 << 30;
    {
        const int per = (n + parts - 1) / parts;
        max_pairs = env_pairs ? env_pairs : per < PRE_MIN_CHUNK ? PRE_MIN_CHUNK : per > 16384 ? 16384 : per;
    }
    P->max_pairs = max_pairs; P->max_bytes = max_bytes;
    P->threaded = next_pchunk(jobs, n, 0, max_pairs, max_bytes, PRE_MIN_CHUNK) < n;
    F->X = X; F->self = P; F->nstage = 4; F->threaded = P->threaded;
    F->cut = p_cut; F->stage[0] = p_send; F->stage[1] = p_l1; F->stage[2] = p_l2; F->stage[3] = p_collect; F->finish = p_finish;
    if (P->threaded && mzi_flow_streams(X)) { P->threaded = F->threaded = 0; }
    P->slots = 2;
    {                                                        /* few long merges: four chunks' DPs side by side (mz_batch.c, mz_flow.c: mzi_flow_wide) */
        static int wide_on = -1;
        if (wide_on < 0) { const char *e = getenv("MZ_WIDE"); wide_on = !(e && e[0] == '0'); }
        if (P->threaded && wide_on && n <= 4096 && total_bytes / (size_t)n >= ((size_t)256 << 10) && mzi_flow_wide(X) == 0) P->slots = MZ_QS;      /* (long: a quarter of a megabyte of input a pair and more) */
    }
    rc = mzi_flow_run(F);
    if (rc < 0) mzi_flow_sync(X);
    for (s = 0; s < MZ_SETS; ++s) { free(P->ck[s].a); free(P->ck[s].where); }
    if (stats) { stats[0] += P->cells; stats[1] += P->bytes_up; stats[2] += P->bytes_down; }
    if (g_ptiming >= 2 && X->ptime_ready) hipEventDestroy(P->ev0);
    pthread_mutex_destroy(&P->mu);
    free(P); free(F);
    return rc;
}

/* ------------------------------------------------------------------------------------------------ entry point */

typedef struct pre_task { mz_ctx *X; int n, rc; const mz_prejob *jobs; mz_preout *outs; int64_t st[3]; char err[600]; } pre_task;
static void *pre_worker(void *arg)
{
    pre_task *t = (pre_task *)arg;
    t->rc = mzi_pre_on_ctx(t->X, t->n, t->jobs, t->outs, t->st);
    if (t->rc < 0) snprintf(t->err, sizeof t->err, "GPU %d: %s", t->X->device, mz_last_error());
    return NULL;
}

static int64_t g_pre_last[3];

int mz_preyama_batch(int n, const mz_prejob *jobs, mz_preout *outs)
{
    int failed = 0, a, b, use, rc, p;
    int64_t st[3] = { 0, 0, 0 };
    int any0 = 0;
    const double t_call = mzi_now_s();
    if (n <= 0) return 0;
    if (!jobs || !outs) return mzi_set_err("mz_preyama_batch: NULL jobs or outs");
    if (g_ptiming < 0) g_ptiming = mzi_timing();
    for (p = 0; p < n; ++p) {
        const mz_prejob *j = &jobs[p];
        if (j->K < 1 || j->L1 < 1 || j->M_all < 1 || j->N_all < 1 || !j->rows1 || !j->rows2)
            return mzi_set_err("mz_preyama_batch: job %d has an empty block or slice", p);
        if (j->v != 0 && j->v != 1) return mzi_set_err("mz_preyama_batch: job %d: v = %d (0 or 1)", p, j->v);
        if (text_bytes(j) >= ((size_t)1 << 31)) return mzi_set_err("mz_preyama_batch: job %d: the two slices hold %zu bytes (2^31 at most)", p, text_bytes(j));
        if (j->v == 0) any0 = 1;
    }
    pthread_mutex_lock(&g_big);
    mzi_link_forget();
    if (mzi_ensure_init() || mzi_sync_scores()) { pthread_mutex_unlock(&g_big); return -1; }
    for (a = 0; a < 128; ++a)                              /* k_fin's pair sums need ss[x][y] == ss[y][x] */
        for (b = 0; b < a; ++b)
            if (ss[a][b] != ss[b][a]) { pthread_mutex_unlock(&g_big); mzi_set_err("score table is not symmetric"); return -2; }
    for (p = 0; p < n; ++p) { memset(&outs[p], 0, sizeof outs[p]); outs[p].status = MZ_E_DEVICE; }
    use = g_ndev;
    while (use > 1 && n / use < MZ_MULTI_MIN) --use;
    if (use == 1) {
        rc = mzi_pre_on_ctx(&G, n, jobs, outs, st);
    } else {
        /* several GPUs: dealt by text volume in a snake over the GPUs (mzi_deal_snake, as mz_yama_batch: every GPU gets the same mix of
         * large and small merges), each GPU's jobs in a list of its own, one host thread per GPU, results back at the jobs' places */
        pre_task task[MZ_MAX_DEV];
        pthread_t th[MZ_MAX_DEV];
        int d, started[MZ_MAX_DEV], cnt[MZ_MAX_DEV], first[MZ_MAX_DEV];
        int *owner = (int *)malloc((size_t)n * sizeof *owner), *where = (int *)malloc((size_t)n * sizeof *where);
        double *wt = (double *)malloc((size_t)n * sizeof *wt);
        mz_prejob *jbuf = (mz_prejob *)malloc((size_t)n * sizeof *jbuf);
        mz_preout *obuf = (mz_preout *)malloc((size_t)n * sizeof *obuf);
        if (!owner || !where || !wt || !jbuf || !obuf) {
            free(owner); free(where); free(wt); free(jbuf); free(obuf);
            pthread_mutex_unlock(&g_big);
            return mzi_set_err("out of memory");
        }
        for (p = 0; p < n; ++p) wt[p] = (double)text_bytes(&jobs[p]) * (jobs[p].v == 0 ? 2.0 : 1.0);
        if (mzi_deal_snake(n, wt, use, owner, where, cnt, first)) { free(owner); free(where); free(wt); free(jbuf); free(obuf); pthread_mutex_unlock(&g_big); return -1; }
        free(wt);
        for (d = 0; d < use; ++d) {
            task[d].X = &g_dev[d]; task[d].jobs = jbuf + first[d]; task[d].outs = obuf + first[d];
            task[d].n = cnt[d]; task[d].rc = 0; task[d].err[0] = 0; memset(task[d].st, 0, sizeof task[d].st);
        }
        for (p = 0; p < n; ++p) { jbuf[where[p]] = jobs[p]; obuf[where[p]] = outs[p]; }
        for (d = 1; d < use; ++d) {
            started[d] = task[d].n > 0 && pthread_create(&th[d], NULL, pre_worker, &task[d]) == 0;
            if (!started[d] && task[d].n > 0) pre_worker(&task[d]);
        }
        if (task[0].n > 0) pre_worker(&task[0]);
        rc = 0;
        for (d = 0; d < use; ++d) {
            if (d >= 1 && started[d]) pthread_join(th[d], NULL);
            if (task[d].n <= 0) continue;
            if (task[d].rc < 0) { rc = -1; mzi_set_err("%s", task[d].err); }
            else failed += task[d].rc;
            st[0] += task[d].st[0]; st[1] += task[d].st[1]; st[2] += task[d].st[2];
        }
        for (p = 0; p < n; ++p) outs[p] = obuf[where[p]];    /* (a chunk's result block hangs on its first merge: mz_free_preouts() walks all n) */
        free(owner); free(where); free(jbuf); free(obuf);
        hipSetDevice(G.device);
        if (rc >= 0) rc = failed;
    }
    g_pre_last[0] = st[0]; g_pre_last[1] = st[1]; g_pre_last[2] = st[2];
    pthread_mutex_unlock(&g_big);
    if (g_ptiming && rc >= 0) {
        const double dt = mzi_now_s() - t_call;
        fprintf(stderr, "{\"mz_preyama_batch\": {\"merges\": %d, \"two_stage\": %d, \"without_block\": %d, \"cells\": %lld, \"seconds\": %.6f, \"gcups\": %.2f, "
                        "\"bytes_up\": %lld, \"bytes_down\": %lld, \"gpus\": %d}}\n",
                n, any0, rc, (long long)st[0], dt, (double)st[0] / dt / 1e9, (long long)st[1], (long long)st[2], use);
    }
    return rc;
}

void mz_pre_link_bytes(int64_t *up, int64_t *down, int64_t *cells)
{
    if (cells) *cells = g_pre_last[0];
    if (up) *up = g_pre_last[1];
    if (down) *down = g_pre_last[2];
}

void mz_free_preouts(int n, mz_preout *outs)
{
    int p;
    for (p = 0; p < n; ++p) { mzi_block_put(outs[p].block); outs[p].block = NULL; outs[p].rows = NULL; outs[p].size = NULL; }
}
